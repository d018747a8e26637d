// kb_vanilla_split16.hip -- Vanilla.Update with one filter split over eight lanes (kb_vanilla_split.h): 13..16 states
// (any p <= 8, m <= 2): two rows per lane, eight filters per wave, 17 KB of LDS per wave (two waves per SIMD).
#include "kb_vanilla_split.h"

namespace kb {

// exact 16 / 8 / 0, Noiseless: the corner of the eight-lane envelope without the run-time dimensions (a third fewer instructions)
template <typename T, int NS, int NM, int NC, int L>
static bool split_exact(const Batch &b, const StepArgs &a) {
    if (a.n != NS || a.p != NM || (a.need_ctrl ? a.m : 0) != NC || a.noise_kind != KB_NOISE_NOISELESS || a.predict) return false;   // (pure predictors: the run-time-everything kernel)
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    const dim3 grid((unsigned)(a.ntiles * L)), block(64);
#define KB_GO(F_, P_) KB_LAUNCH((vanilla_split_kernel<T, NS, NM, NC, L, false, F_, P_>), grid, block, 0, b.stream, a)
    if (full) KB_GO(true, false); else KB_GO(false, false);
#undef KB_GO
    return true;
}

bool launch_vanilla_split16(const Batch &b, const StepArgs &a) {
    if (b.dtype != KB_F64 || a.n > 16 || a.p > 8 || (a.need_ctrl ? a.m : 0) > 2) return false;
    // (BatchNoise runs here since round 6: kb_vanilla_split12.hip)
    if (split_exact<double, 16, 8, 0, 8>(b, a)) return true;
    if (launch_vanilla_split16_plain(b, a)) return true;
    KB_LAUNCH((vanilla_split_kernel<double, 16, 8, 2, 8, true, false, false>), dim3((unsigned)(a.ntiles * 8)), dim3(64), 0, b.stream, a);
    return true;
}

}  // namespace kb
