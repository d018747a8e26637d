// kb_srif_pair64b.hip -- more shapes of the two-lanes-per-filter SRIF Update (kb_srif_pair.h), fp64: 6, 8 and 10 states with 1 to 4
// measurements (an odd number of measurements on the next even instantiation: kb_srif_pair.h PADM) (orbit-determination filters with estimated parameters beside the six orbital states).
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f64b(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<double, 8, 2, true>(b, a) || srif_pair_launch<double, 8, 4, true>(b, a) || srif_pair_launch<double, 10, 2, true>(b, a) || srif_pair_launch<double, 10, 4, true>(b, a) ||
           srif_pair_launch<double, 6, 4, true>(b, a) || (a.p == 1 && srif_pair_launch<double, 6, 2, true>(b, a));   // (6 / 2 itself: the exact kernel of kb_srif_pair64.hip)
}
}  // namespace kb
