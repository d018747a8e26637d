// kb_vanilla_shapes.hip -- register-resident Vanilla kernels (kb_vanilla_reg.h) for the shapes the
// reference's own examples and tests use, fp64, one step per launch:
//   examples/robot (n=2, p=1, m=1), Midterm2 tests (3/1/1), examples/jerkcar and the *MultiD tests
//   (4/1/1 and 4/2/1: H is swapped between a 1-row and a 2-row matrix), examples/statOD5044 (4/2/2),
//   hybrid-sized 6/2.
#include "kb_vanilla_reg.h"

namespace kb {

bool launch_vanilla_extra_shapes(const Batch &b, const StepArgs &a, bool fused) {
    if (fused) return false;  // the time-fused variant is only built for the benchmark shapes
    return try_reg<double, 4, 1, 1, false>(b, a, false) || try_reg<double, 4, 2, 1, false>(b, a, false) || try_reg<double, 2, 1, 1, false>(b, a, false) ||
           try_reg<double, 2, 1, 0, false>(b, a, false) || try_reg<double, 3, 1, 1, false>(b, a, false) || try_reg<double, 3, 1, 0, false>(b, a, false) ||
           try_reg<double, 4, 2, 2, false>(b, a, false) || try_reg<double, 4, 1, 0, false>(b, a, false) || try_reg<double, 6, 2, 0, false>(b, a, false);
}

}  // namespace kb
