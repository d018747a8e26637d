// kb_srif_pair32d.hip -- more shapes of the two-lanes-per-filter SRIF Update (kb_srif_pair.h), fp32: 6, 8 and 10 states with 5 or 6
// measurements (5 on the six-row instantiation with one padded row: kb_srif_pair.h PADM).
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f32d(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<float, 6, 6, true>(b, a) || srif_pair_launch<float, 8, 6, true>(b, a) || srif_pair_launch<float, 10, 6, true>(b, a);
}
}  // namespace kb
