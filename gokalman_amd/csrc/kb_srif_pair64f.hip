// kb_srif_pair64f.hip -- the two-lanes-per-filter SRIF Update (kb_srif_pair.h) at 14 states, fp64: 1 to 4 measurements (5, 6: kb_srif_pair64f6.hip) (odd counts on the next even
// instantiation, PADM).  One wave per SIMD; the panel does not fit the register file in fp64 (14 states: 32-616 B, 16 states: 1-1.9 KB of
// scratch per lane) -- still an order of magnitude under the statement kernel's 10-31 KB: 14/4 475 us, 16/4 1.25 ms per 256k-filter step
// against 13.9 / 20.2 ms.  Predict() at these sizes stays on the statement kernel.  (One translation unit per row count: each takes
// minutes to compile.)
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f64f6(const Batch &b, const StepArgs &a);   // kb_srif_pair64f6.hip
bool launch_srif_pair_f64f(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<double, 14, 2, true>(b, a) || srif_pair_launch<double, 14, 4, true>(b, a) || launch_srif_pair_f64f6(b, a);
}
}  // namespace kb
