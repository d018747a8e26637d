// kb_hybrid_pad8.hip -- the padded HybridKF update for 7 and 8 states (and p = 4 at 5, 6 states falls to kb_hybrid_pad.hip): any
// n <= 8, p <= 4; a translation unit of its own so that the build compiles the 16 heavy instantiations beside the others.
#include "kb_hybrid_reg.h"

namespace kb {

bool launch_hybrid_padded8(const Batch &b, const StepArgs &a) {
    return hybrid_try<double, 8, 4, true>(b, a);
}

}  // namespace kb
