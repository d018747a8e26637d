// kb_hybrid_reg.hip -- register-resident HybridKF update (hybrid.go:104-204) for the statOD-sized
// ensemble (n = 6, p = 2): CKF or EKF, per-step Phi and H-tilde streamed from HBM (non-temporal),
// x and packed P read and rewritten, SNC (PreparePNT, q <= 3) and Predict() as wave-uniform branches.
// Algorithmic bytes per filter-step: x 6 + P 36 + Phi 36 + Htilde 12 + R 4 + real 2 + computed 2
// read, x 6 + P 36 written = 1120 B.
#include "kb_hybrid_reg.h"

namespace kb {


bool hybrid_reg_ok(const Batch &b, const StepArgs &a) {
    return b.dtype == KB_F64 && (hybrid_shape_ok(a, 6, 2) || hybrid_shape_ok(a, 6, 3) || hybrid_shape_ok(a, 6, 1) || hybrid_shape_ok(a, 8, 4, true) || hybrid_split_ok(b, a));
}

int launch_hybrid(const Batch &b, const StepArgs &a) {
    bool done = false;
    if (b.dtype == KB_F64) done = hybrid_try<double, 6, 2>(b, a) || hybrid_try<double, 6, 3>(b, a) || hybrid_try<double, 6, 1>(b, a) ||
                                  launch_hybrid_padded(b, a) || launch_hybrid_padded8(b, a);   // kb_hybrid_pad.hip, kb_hybrid_pad8.hip: any n <= 8, p <= 4
    if (!done) done = launch_hybrid_split(b, a);   // kb_hybrid_split.hip: everything else up to 16 / 8
    if (!done && (a.flags & KB_FLAG_STRICT_SYMCHECK) && !(a.flags & KB_FLAG_STATEMENT_KERNELS)) done = launch_hybrid_strict(b, a);   // kb_hybrid_strict.hip
    if (!done) return launch_hybrid_gen(b, a);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
