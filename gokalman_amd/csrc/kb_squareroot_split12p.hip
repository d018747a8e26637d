// kb_squareroot_split12p.hip -- SquareRoot.Update, one filter over four lanes (kb_squareroot_split.h), the PADDED shapes of the
// common case: Noiseless or AWGN (FULL estimates and AWGN up to p = 6), m <= 2 -- n <= 8 with p <= 4 on an 8-state instantiation (two columns per lane), n <= 12
// with p <= 4 / 6 / 8 on 12-state ones; run-time dimensions on the exact kernel's schedule (RT = false).
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

bool launch_squareroot_split12_plain(const Batch &b, const StepArgs &a) {
    const int m = a.need_ctrl ? a.m : 0;
    if (b.dtype != KB_F64 || a.n > 12 || a.p > 8 || m > 2 || a.sqrt_p != a.p || a.nsteps != 1) return false;
    if (a.noise_kind != KB_NOISE_NOISELESS && a.noise_kind != KB_NOISE_AWGN) return false;
    if (((a.flags & KB_FLAG_FULL_ESTIMATE) || a.noise_kind == KB_NOISE_AWGN) && a.p > 6) return false;   // (FULL or AWGN with 7 or 8 measurements: the run-time-everything kernel)
    if (a.n <= 8 && a.p <= 4) sq_plain<8, 4, 4>(b, a);
    else if (a.p <= 4) sq_plain<12, 4, 4>(b, a);
    else if (a.p <= 6) sq_plain<12, 6, 4>(b, a);
    else sq_plain<12, 8, 4>(b, a);
    return true;
}

}  // namespace kb
