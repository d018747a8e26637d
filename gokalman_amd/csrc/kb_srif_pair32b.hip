// kb_srif_pair32b.hip -- more shapes of the two-lanes-per-filter SRIF Update (kb_srif_pair.h), fp32: 8 and 10 states with 2 or 4
// measurements (orbit-determination filters with estimated parameters beside the six orbital states).
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f32b(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<float, 8, 2>(b, a) || srif_pair_launch<float, 8, 4>(b, a) || srif_pair_launch<float, 10, 2>(b, a) || srif_pair_launch<float, 10, 4>(b, a);
}
}  // namespace kb
