// kb_srif_pair32c.hip -- more shapes of the two-lanes-per-filter SRIF Update (kb_srif_pair.h), fp32: 12 states with 1 to 5 measurements (odd counts: kb_srif_pair.h PADM).
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f32c(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<float, 12, 2, true>(b, a) || srif_pair_launch<float, 12, 4, true>(b, a) || srif_pair_launch<float, 12, 6, true>(b, a);
}
}  // namespace kb
