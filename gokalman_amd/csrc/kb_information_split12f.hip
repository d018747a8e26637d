// kb_information_split12f.hip -- Information.Update with KB_FLAG_FULL_ESTIMATE (yhat = H State(prev), I- for the Estimate), one filter
// split over four (n <= 12) / eight (n <= 16) lanes (kb_information_split.h FULLT): Noiseless and AWGN.
#include "kb_information_split.h"

namespace kb {

bool launch_information_split_full(const Batch &b, const StepArgs &a) {
    const int m = a.need_ctrl ? a.m : 0;
    if (a.n <= 8 && a.p <= 4)
        KB_LAUNCH((information_split_kernel<double, 8, 4, 2, 4, true, true>), dim3((unsigned)(a.ntiles * 4)), dim3(64), 0, b.stream, a);
    else if (a.n == 12 && a.p == 6 && m == 0)
        KB_LAUNCH((information_split_kernel<double, 12, 6, 0, 4, false, true>), dim3((unsigned)(a.ntiles * 4)), dim3(64), 0, b.stream, a);
    else if (a.n <= 12)
        KB_LAUNCH((information_split_kernel<double, 12, 8, 2, 4, true, true>), dim3((unsigned)(a.ntiles * 4)), dim3(64), 0, b.stream, a);
    else
        KB_LAUNCH((information_split_kernel<double, 16, 8, 2, 8, true, true>), dim3((unsigned)(a.ntiles * 8)), dim3(64), 0, b.stream, a);
    return true;
}

}  // namespace kb
