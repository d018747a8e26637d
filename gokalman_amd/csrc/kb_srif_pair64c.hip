// kb_srif_pair64c.hip -- more shapes of the two-lanes-per-filter SRIF Update (kb_srif_pair.h), fp64: 12 states with 2 or 4 measurements.
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f64c(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<double, 12, 2>(b, a) || srif_pair_launch<double, 12, 4>(b, a);
}
}  // namespace kb
