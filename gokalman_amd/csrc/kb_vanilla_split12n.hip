// kb_vanilla_split12n.hip -- Vanilla.Update on four lanes per filter (kb_vanilla_split.h) at the benchmark shape 12 / 6 / 0 WITH a Noise:
// exact dimensions, the Noise kind a compile-time constant (NOISET: 1 AWGN, 2 BatchNoise), state-only and KB_FLAG_FULL_ESTIMATE.
// Round 6 (VERDICT r05 task 5): BatchNoise beyond 8 states used to run the statement-order kernel (8.7 ms per 256k-filter step),
// AWGN at this shape the run-time-everything kernel <12, 8, 2, GEN>.  Pure predictors stay on the latter.
#include "kb_vanilla_split.h"

namespace kb {

bool launch_vanilla_split12_noise(const Batch &b, const StepArgs &a) {
    if (b.dtype != KB_F64 || a.n != 12 || a.p != 6 || a.need_ctrl || a.predict || a.noise_kind == KB_NOISE_NOISELESS) return false;
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    const dim3 grid((unsigned)(a.ntiles * 4)), block(64);
#define KB_GO(F_, N_) KB_LAUNCH((vanilla_split_kernel<double, 12, 6, 0, 4, false, F_, false, false, false, N_>), grid, block, 0, b.stream, a)
    if (a.noise_kind == KB_NOISE_BATCH) { if (full) KB_GO(true, 2); else KB_GO(false, 2); }
    else                                { if (full) KB_GO(true, 1); else KB_GO(false, 1); }
#undef KB_GO
    return true;
}

}  // namespace kb
