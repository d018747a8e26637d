// kb_vanilla_split12p.hip -- Vanilla.Update, one filter over four lanes (kb_vanilla_split.h), the PADDED shapes of the common case:
// any n <= 12, m <= 2 with p <= 4 / 6 / 8, Noiseless or AWGN, with or without FULL estimates, no pure prediction -- run-time dimensions on the exact
// kernel's schedule (RT = false).  Everything else with n <= 12 stays on the run-time-everything kernel (kb_vanilla_split12.hip).
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

bool launch_vanilla_split12_plain(const Batch &b, const StepArgs &a) {
    if (b.dtype != KB_F64 || a.n > 12 || a.p > 8 || (a.need_ctrl ? a.m : 0) > 2) return false;
    if ((a.noise_kind != KB_NOISE_NOISELESS && a.noise_kind != KB_NOISE_AWGN) || a.predict) return false;
    if (a.noise_kind == KB_NOISE_AWGN && (a.flags & KB_FLAG_FULL_ESTIMATE) && a.p > 6) return false;   // (AWGN with FULL at 7, 8 measurements: the run-time-everything kernel)
    if (a.p <= 4) split_plain<12, 4, 4>(b, a);
    else if (a.p <= 6) split_plain<12, 6, 4>(b, a);
    else split_plain<12, 8, 4>(b, a);
    return true;
}

}  // namespace kb
