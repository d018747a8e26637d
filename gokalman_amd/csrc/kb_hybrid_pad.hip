// kb_hybrid_pad.hip -- the padded family of the register-resident HybridKF update (kb_hybrid_reg.h PAD): any n <= 4 with p <= 2, any
// n <= 6 with p <= 4 that has no exact instantiation (CKF / EKF, FULL, zero-copy Phi / Htilde, SNC, Predict as the exact kernels).
#include "kb_hybrid_reg.h"

namespace kb {

bool launch_hybrid_padded(const Batch &b, const StepArgs &a) {
    return hybrid_try<double, 4, 2, true>(b, a) || hybrid_try<double, 6, 4, true>(b, a);
}

}  // namespace kb
