// kb_srif_pair32f.hip -- the two-lanes-per-filter SRIF Update (kb_srif_pair.h) at 14 states, fp32: 1 to 6 measurements (odd counts on the
// next even instantiation, PADM).  One wave per SIMD; in fp64 the 16-state panel does not fit the register file (1.3-1.6 KB of scratch per
// lane) -- still an order of magnitude under the statement kernel's 10-31 KB.  Predict() at these sizes stays on the statement kernel.
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f32f(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<float, 14, 2, true>(b, a) || srif_pair_launch<float, 14, 4, true>(b, a) || srif_pair_launch<float, 14, 6, true>(b, a);
}
}  // namespace kb
