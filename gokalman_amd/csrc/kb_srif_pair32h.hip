// kb_srif_pair32h.hip -- SRIF Update, two lanes per filter (kb_srif_pair.h), TIME-FUSED: config E's shape (12 / 6, fp32, zero-copy Phi /
// Htilde / observations, steady state, state only) with the caller loop inside one launch (kb_update_nl_steps_dev, round 6).
#include "kb_srif_pair.h"

namespace kb {

bool launch_srif_pair_f32_fused(const Batch &b, const StepArgs &a) {
    if (b.dtype != KB_F32) return false;
    return srif_pair_launch_fused<float, 12, 6>(b, a) || srif_pair_launch_fused<float, 6, 2>(b, a);   // (6 / 2: the reference's own SRIF shape, srif_test.go / examples/statOD)
}

}  // namespace kb
