// kb_hybrid_split.hip -- HybridKF.fullUpdate (hybrid.go:104-204) beyond 8 states or 4 measurements (what the register kernels do not take): the split-lane Vanilla kernel (kb_vanilla_split.h,
// one filter over four / eight lanes) in its HYB mode -- CKF or EKF (StepArgs::ekf), Phi / Htilde from the model block (kb_prepare) or in place from
// the caller's planar arrays (kb_prepare_dev: zero copy), R from the model block, SNC (PreparePNT, q <= 3), Predict(), p <= 6,
// with and without KB_FLAG_FULL_ESTIMATE; p = 7, 8 in kb_hybrid_split8.hip (the same template).  SNC with q > 3 and the strict symmetry
// test stay on hybrid_gen_kernel.
#include "kb_vanilla_split.h"

namespace kb {

bool launch_hybrid_split8(const Batch &b, const StepArgs &a);   // kb_hybrid_split8.hip

template <int NS, int NM, int L>
static void hyb_go(const Batch &b, const StepArgs &a) {
    const dim3 grid((unsigned)(a.ntiles * L)), block(64);
    if (a.predict) {   // Predict() (hybrid.go:125-143): {xBar (CKF) or 0 (EKF), PBar}; the six-measurement instantiations carry any p <= 6
        if constexpr (NM == 6) {
            if (a.flags & KB_FLAG_FULL_ESTIMATE) KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 0, L, true, true, true, false, false, false, true>), grid, block, 0, b.stream, a);
            else KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 0, L, true, false, true, false, false, false, true>), grid, block, 0, b.stream, a);
        }
        return;
    }
    if (a.flags & KB_FLAG_FULL_ESTIMATE) KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 0, L, true, true, false, false, false, false, true>), grid, block, 0, b.stream, a);
    else KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 0, L, true, false, false, false, false, false, true>), grid, block, 0, b.stream, a);
}

bool hybrid_split_ok(const Batch &b, const StepArgs &a) {
    if (b.dtype != KB_F64 || a.n > 16 || a.p > 8 || (a.snc && a.L.nq > 3)) return false;
    return !(a.flags & (KB_FLAG_STRICT_SYMCHECK | KB_FLAG_STATEMENT_KERNELS));
}

bool launch_hybrid_split(const Batch &b, const StepArgs &a) {
    if (!hybrid_split_ok(b, a)) return false;
    if (a.p > 6) return launch_hybrid_split8(b, a);
    if (a.n <= 12) { if (a.p <= 4 && !a.predict) hyb_go<12, 4, 4>(b, a); else hyb_go<12, 6, 4>(b, a); }
    else           { if (a.p <= 4 && !a.predict) hyb_go<16, 4, 8>(b, a); else hyb_go<16, 6, 8>(b, a); }
    return true;
}

}  // namespace kb
