// kb_hybrid_split8.hip -- kb_hybrid_split.hip's kernels for seven and eight measurements (8 < n <= 16), Update and Predict, with and without
// KB_FLAG_FULL_ESTIMATE: the eight-measurement instantiations of kb_vanilla_split.h in its HYB mode (one wave per SIMD at four lanes per filter).
#include "kb_vanilla_split.h"

namespace kb {

template <int NS, int NM, int L>
static void hyb_go(const Batch &b, const StepArgs &a) {
    const dim3 grid((unsigned)(a.ntiles * L)), block(64);
    if (a.predict) {   // Predict() (hybrid.go:125-143): {xBar (CKF) or 0 (EKF), PBar}; the six-measurement instantiations carry any p <= 6
        if constexpr (NM >= 6) {
            if (a.flags & KB_FLAG_FULL_ESTIMATE) KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 0, L, true, true, true, false, false, false, true>), grid, block, 0, b.stream, a);
            else KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 0, L, true, false, true, false, false, false, true>), grid, block, 0, b.stream, a);
        }
        return;
    }
    if (a.flags & KB_FLAG_FULL_ESTIMATE) KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 0, L, true, true, false, false, false, false, true>), grid, block, 0, b.stream, a);
    else KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 0, L, true, false, false, false, false, false, true>), grid, block, 0, b.stream, a);
}

bool launch_hybrid_split8(const Batch &b, const StepArgs &a) {
    if (a.n <= 12) hyb_go<12, 8, 4>(b, a);
    else hyb_go<16, 8, 8>(b, a);
    return true;
}

}  // namespace kb
