// kb_srif_pair32.hip -- fp32 instantiations of the two-lanes-per-filter SRIF Update (kb_srif_pair.h); config E is 12/6 fp32.
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f32(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<float, 12, 6>(b, a) || srif_pair_launch<float, 6, 2>(b, a);
}
}  // namespace kb
