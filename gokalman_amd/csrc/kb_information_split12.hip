// kb_information_split12.hip -- Information.Update with one filter split over four (n <= 12) / eight (n <= 16) lanes
// (kb_information_split.h); the instantiations with KB_FLAG_FULL_ESTIMATE: kb_information_split12f.hip.
#include "kb_information_split.h"

namespace kb {

bool launch_information_split(const Batch &b, const StepArgs &a) {
    const int m = a.need_ctrl ? a.m : 0;
    if (b.dtype != KB_F64 || a.n > 16 || a.p > 8 || m > 2 || a.rinv_p != a.p || a.nsteps != 1) return false;
    if (a.flags & KB_FLAG_STRICT_SYMCHECK) return false;
    if (a.noise_kind == KB_NOISE_BATCH) return false;
    if (a.flags & KB_FLAG_FULL_ESTIMATE) return launch_information_split_full(b, a);   // kb_information_split12f.hip
    if (launch_information_split8(b, a)) return true;
    if (a.n == 12 && a.p == 6 && m == 0)
        KB_LAUNCH((information_split_kernel<double, 12, 6, 0, 4, false>), dim3((unsigned)(a.ntiles * 4)), dim3(64), 0, b.stream, a);
    else if (a.n <= 12)
        KB_LAUNCH((information_split_kernel<double, 12, 8, 2, 4, true>), dim3((unsigned)(a.ntiles * 4)), dim3(64), 0, b.stream, a);
    else
        KB_LAUNCH((information_split_kernel<double, 16, 8, 2, 8, true>), dim3((unsigned)(a.ntiles * 8)), dim3(64), 0, b.stream, a);
    return true;
}

}  // namespace kb
