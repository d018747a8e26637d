// kb_srif_pair64.hip -- fp64 instantiations of the two-lanes-per-filter SRIF Update (kb_srif_pair.h).
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f64(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<double, 12, 6>(b, a) || srif_pair_launch<double, 6, 2>(b, a);
}
}  // namespace kb
