// kb_vanilla_split12.hip -- Vanilla.Update with one filter split over four lanes (kb_vanilla_split.h): 12 states.
//   exact 12 / 6 / 0, Noiseless: the orbit-determination-sized shape the SRIF benchmark (config E) uses, per-filter models
//   padded shapes, Noiseless, state only: kb_vanilla_split12p.hip
//   exact 12 / 6 / 0 with run-time FULL / PREDICT / Noise (AWGN, BatchNoise)
//   GEN <12, 8, 2> with run-time FULL / PREDICT / Noise: every other batch with n <= 12, p <= 8, m <= 2 that has no one-filter-per-lane register kernel
#include "kb_vanilla_split.h"

namespace kb {

template <typename T, int NS, int NM, int NC, int L, bool WITH_PREDICT = true>
static bool split_exact(const Batch &b, const StepArgs &a) {
    if (a.n != NS || a.p != NM || (a.need_ctrl ? a.m : 0) != NC || a.noise_kind != KB_NOISE_NOISELESS) return false;
    if (!WITH_PREDICT && a.predict) return false;   // (pure predictors of this shape: the run-time-everything kernel)
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
#ifdef KB_SPLIT_PERSIST
    const int64_t slots = (int64_t)KB_SPLIT_PERSIST;
    const dim3 grid((unsigned)(a.ntiles * L < slots ? a.ntiles * L : slots)), block(64);
#define KB_GO(F_, P_) KB_LAUNCH((vanilla_split_kernel<T, NS, NM, NC, L, false, F_, P_, true>), grid, block, 0, b.stream, a)
#else
    const dim3 grid((unsigned)(a.ntiles * L)), block(64);
#define KB_GO(F_, P_) KB_LAUNCH((vanilla_split_kernel<T, NS, NM, NC, L, false, F_, P_>), grid, block, 0, b.stream, a)
#endif
    if (a.predict) { if constexpr (WITH_PREDICT) { if (full) KB_GO(true, true); else KB_GO(false, true); } }
    else           { if (full) KB_GO(true, false); else KB_GO(false, false); }
#undef KB_GO
    return true;
}

bool launch_vanilla_split12(const Batch &b, const StepArgs &a) {
    if (b.dtype != KB_F64 || a.n > 12 || a.p > 8 || (a.need_ctrl ? a.m : 0) > 2) return false;
    // BatchNoise (noise.go:67-106, zero Q / R) runs HERE since round 6 (until then: the statement-order kernel, 45x slower).  Measured in
    // round 5 and settled by the 60-digit arbiter in round 6 (tests/golden/make_highprec.py): up to n measurements every evaluation order is
    // exact to rounding, and past them the exact S is SINGULAR -- no order of operations has digits to keep there.
    if (launch_vanilla_split12_noise(b, a)) return true;   // kb_vanilla_split12n.hip: the benchmark shape 12 / 6 with AWGN / BatchNoise
    if (split_exact<double, 12, 6, 0, 4>(b, a)) return true;
    if (split_exact<double, 12, 8, 0, 4, false>(b, a)) return true;   // (the corner of the four-lane envelope)
    if (launch_vanilla_split12_plain(b, a)) return true;
    KB_LAUNCH((vanilla_split_kernel<double, 12, 8, 2, 4, true, false, false>), dim3((unsigned)(a.ntiles * 4)), dim3(64), 0, b.stream, a);
    return true;
}

}  // namespace kb
