// kb_vanilla_shared.hip -- the Vanilla register kernels (kb_vanilla_reg.h) instantiated for batches in which every filter has
// the SAME model (F, G, H, Q, R uploaded with broadcast = 1: one gokalman filter object fanned out over many runs or targets --
// kb_replicate, BatchLDKF, the chi-square ensembles).  StepArgs::mo_ts is 0 for such a batch: every wave reads tile 0's model
// block, with the default cache policy (SHARED), so the 43 KB stay in each XCD's L2 and only state and measurements cross the
// fabric: 432 instead of 1104 bytes per filter-step at 6/3.  1M filters: 105 us per step against 166 us with per-filter models.
// fp64, one step per launch; the exact benchmark shapes and the padded families (any n <= 8, p <= 4, m <= 2).
#include "kb_vanilla_reg.h"

namespace kb {

bool launch_vanilla_shared(const Batch &b, const StepArgs &a) {
    if (a.noise_kind == KB_NOISE_NOISELESS)
        return try_reg<double, 6, 3, 0, false, false, true>(b, a, false) || try_reg<double, 4, 2, 0, false, false, true>(b, a, false) ||
               try_pad<double, 4, 2, 0, false, true>(b, a) || try_pad<double, 4, 2, 2, false, true>(b, a) ||
               try_pad<double, 6, 4, 0, false, true>(b, a) || try_pad<double, 6, 4, 2, false, true>(b, a) ||
               try_pad<double, 8, 4, 0, false, true>(b, a) || try_pad<double, 8, 4, 2, false, true>(b, a);
    return try_reg<double, 6, 3, 0, false, true, true>(b, a, false) || try_pad<double, 4, 2, 0, true, true>(b, a) || try_pad<double, 4, 2, 2, true, true>(b, a) ||
           try_pad<double, 6, 4, 0, true, true>(b, a) || try_pad<double, 6, 4, 2, true, true>(b, a) ||
           try_pad<double, 8, 4, 0, true, true>(b, a) || try_pad<double, 8, 4, 2, true, true>(b, a);
}

}  // namespace kb
