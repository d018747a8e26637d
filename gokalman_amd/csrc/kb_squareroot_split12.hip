// kb_squareroot_split12.hip -- SquareRoot.Update with one filter split over four lanes (kb_squareroot_split.h): up to 12 states.
#include "kb_squareroot_split.h"

namespace kb {

bool launch_squareroot_split12(const Batch &b, const StepArgs &a) {
    const int m = a.need_ctrl ? a.m : 0;
    if (b.dtype != KB_F64 || a.n > 12 || a.p > 8 || m > 2 || a.sqrt_p != a.p || a.nsteps != 1) return false;
    if (a.noise_kind != KB_NOISE_NOISELESS && a.noise_kind != KB_NOISE_AWGN) return false;
    const dim3 grid((unsigned)(a.ntiles * 4)), block(64);
    if (a.n == 12 && a.p == 6 && m == 0 && a.noise_kind == KB_NOISE_NOISELESS) {
        if (a.flags & KB_FLAG_FULL_ESTIMATE) KB_LAUNCH((squareroot_split_kernel<double, 12, 6, 0, 4, false, true>), grid, block, 0, b.stream, a);
        else KB_LAUNCH((squareroot_split_kernel<double, 12, 6, 0, 4, false, false>), grid, block, 0, b.stream, a);
        return true;
    }
    if (launch_squareroot_split12_plain(b, a)) return true;
    KB_LAUNCH((squareroot_split_kernel<double, 12, 8, 2, 4, true, false>), grid, block, 0, b.stream, a);
    return true;
}

}  // namespace kb
