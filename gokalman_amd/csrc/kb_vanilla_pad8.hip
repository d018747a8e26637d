// kb_vanilla_pad8.hip -- the n <= 8 members of the padded Vanilla kernel family (see kb_vanilla_pad.hip); a separate
// translation unit so that the build compiles them in parallel.
#include "kb_vanilla_reg.h"

namespace kb {

bool launch_vanilla_padded8(const Batch &b, const StepArgs &a) {
    return try_pad<double, 8, 2, 0>(b, a) || try_pad<double, 8, 2, 2>(b, a) || try_pad<double, 8, 4, 0>(b, a) || try_pad<double, 8, 4, 2>(b, a);
}

}  // namespace kb
