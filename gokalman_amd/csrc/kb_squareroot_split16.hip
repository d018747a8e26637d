// kb_squareroot_split16.hip -- SquareRoot.Update with one filter split over eight lanes (kb_squareroot_split.h): 13..16 states
// (any p <= 8, m <= 2): two columns of each panel per lane, eight filters per wave.
#include "kb_squareroot_split.h"

namespace kb {

bool launch_squareroot_split16(const Batch &b, const StepArgs &a) {
    const int m = a.need_ctrl ? a.m : 0;
    if (b.dtype != KB_F64 || a.n > 16 || a.p > 8 || m > 2 || a.sqrt_p != a.p || a.nsteps != 1) return false;
    if (a.noise_kind != KB_NOISE_NOISELESS && a.noise_kind != KB_NOISE_AWGN) return false;
    if (launch_squareroot_split16_plain(b, a)) return true;
    KB_LAUNCH((squareroot_split_kernel<double, 16, 8, 2, 8, true, false>), dim3((unsigned)(a.ntiles * 8)), dim3(64), 0, b.stream, a);
    return true;
}

}  // namespace kb
