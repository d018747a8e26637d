// kb_information_split8.hip -- Information.Update, one filter over four lanes (kb_information_split.h) for 7 and 8 states with
// p <= 4, m <= 2: two rows per lane (the one-filter-per-lane kernels end at 6 states).
#include "kb_information_split.h"

namespace kb {

bool launch_information_split8(const Batch &b, const StepArgs &a) {
    if (a.n > 8 || a.p > 4) return false;
    KB_LAUNCH((information_split_kernel<double, 8, 4, 2, 4, true>), dim3((unsigned)(a.ntiles * 4)), dim3(64), 0, b.stream, a);
    return true;
}

}  // namespace kb
