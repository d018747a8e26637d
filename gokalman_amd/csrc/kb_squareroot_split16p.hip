// kb_squareroot_split16p.hip -- SquareRoot.Update, one filter over eight lanes (kb_squareroot_split.h), the padded shapes of the
// common case with 13..16 states: Noiseless or AWGN (FULL estimates and AWGN up to p = 6), m <= 2, p <= 4 / 6 / 8 (see kb_squareroot_split12p.hip).
#include "kb_squareroot_split.h"

namespace kb {

template <int NS, int NM, int L>
static void sq_plain(const Batch &b, const StepArgs &a) {
    const dim3 grid((unsigned)(a.ntiles * L)), block(64);
    if (a.noise_kind == KB_NOISE_AWGN) {   // (p <= 6 only: the caller sends p = 7, 8 with noise to the run-time-everything kernel)
        if constexpr (NM <= 6) {
            if (a.flags & KB_FLAG_FULL_ESTIMATE) KB_LAUNCH((squareroot_split_kernel<double, NS, NM, 2, L, true, true, false, true>), grid, block, 0, b.stream, a);
            else KB_LAUNCH((squareroot_split_kernel<double, NS, NM, 2, L, true, false, false, true>), grid, block, 0, b.stream, a);
        }
    } else if (a.flags & KB_FLAG_FULL_ESTIMATE) {
        if constexpr (NM <= 6) KB_LAUNCH((squareroot_split_kernel<double, NS, NM, 2, L, true, true, false>), grid, block, 0, b.stream, a);
    } else {
        KB_LAUNCH((squareroot_split_kernel<double, NS, NM, 2, L, true, false, false>), grid, block, 0, b.stream, a);
    }
}

bool launch_squareroot_split16_plain(const Batch &b, const StepArgs &a) {
    const int m = a.need_ctrl ? a.m : 0;
    if (b.dtype != KB_F64 || a.n > 16 || a.p > 8 || m > 2 || a.sqrt_p != a.p || a.nsteps != 1) return false;
    if (a.noise_kind != KB_NOISE_NOISELESS && a.noise_kind != KB_NOISE_AWGN) return false;
    if (((a.flags & KB_FLAG_FULL_ESTIMATE) || a.noise_kind == KB_NOISE_AWGN) && a.p > 6) return false;   // (FULL or AWGN with 7 or 8 measurements: the run-time-everything kernel)
    if (a.p <= 4) sq_plain<16, 4, 8>(b, a);
    else if (a.p <= 6) sq_plain<16, 6, 8>(b, a);
    else sq_plain<16, 8, 8>(b, a);
    return true;
}

}  // namespace kb
