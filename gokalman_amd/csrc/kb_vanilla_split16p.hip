// kb_vanilla_split16p.hip -- Vanilla.Update, one filter over eight lanes (kb_vanilla_split.h), the padded shapes of the common case
// with 13..16 states: m <= 2, p <= 4 / 6 / 8, Noiseless or AWGN, with or without FULL estimates, no pure prediction (see kb_vanilla_split12p.hip).
#include "kb_vanilla_split.h"

namespace kb {

template <int NS, int NM, int L>
static void split_plain(const Batch &b, const StepArgs &a) {
    const dim3 grid((unsigned)(a.ntiles * L)), block(64);
    if (a.noise_kind == KB_NOISE_AWGN) {
        if (a.flags & KB_FLAG_FULL_ESTIMATE) { if constexpr (NM <= 6) KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 2, L, true, true, false, false, false, true>), grid, block, 0, b.stream, a); }
        else KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 2, L, true, false, false, false, false, true>), grid, block, 0, b.stream, a);
    } else if (a.flags & KB_FLAG_FULL_ESTIMATE) {
        KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 2, L, true, true, false, false, false>), grid, block, 0, b.stream, a);
    } else {
        KB_LAUNCH((vanilla_split_kernel<double, NS, NM, 2, L, true, false, false, false, false>), grid, block, 0, b.stream, a);
    }
}

bool launch_vanilla_split16_plain(const Batch &b, const StepArgs &a) {
    if (b.dtype != KB_F64 || a.n > 16 || a.p > 8 || (a.need_ctrl ? a.m : 0) > 2) return false;
    if ((a.noise_kind != KB_NOISE_NOISELESS && a.noise_kind != KB_NOISE_AWGN) || a.predict) return false;
    if (a.noise_kind == KB_NOISE_AWGN && (a.flags & KB_FLAG_FULL_ESTIMATE) && a.p > 6) return false;   // (AWGN with FULL at 7, 8 measurements: the run-time-everything kernel)
    if (a.p <= 4) split_plain<16, 4, 8>(b, a);
    else if (a.p <= 6) split_plain<16, 6, 8>(b, a);
    else split_plain<16, 8, 8>(b, a);
    return true;
}

}  // namespace kb
