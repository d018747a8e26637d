// kb_srif_pair32e.hip -- more shapes of the two-lanes-per-filter SRIF Update (kb_srif_pair.h), fp32: 6, 8, 10 and 12 states with 7 or 8
// measurements (7 on the eight-row instantiation with one padded row: kb_srif_pair.h PADM).
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f32e(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<float, 6, 8, true>(b, a) || srif_pair_launch<float, 8, 8, true>(b, a) || srif_pair_launch<float, 10, 8, true>(b, a) ||
           srif_pair_launch<float, 12, 8, true>(b, a);
}
}  // namespace kb
