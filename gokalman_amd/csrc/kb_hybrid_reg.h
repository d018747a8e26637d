// kb_hybrid_reg.h -- the register-resident HybridKF update (hybrid.go:104-204), one filter per lane: the kernel template and its
// launch helper, shared by kb_hybrid_reg.hip (the exact statOD shapes 6/1, 6/2, 6/3) and kb_hybrid_pad*.hip (the padded family: any
// n <= 8, p <= 4).
#pragma once
#include "kb_internal.h"
#include "kb_static.h"

namespace kb {

#ifndef HYB_WPB
#define HYB_WPB 1   // waves per workgroup: a finished wave frees its slot at once (4 per workgroup: 137.7 us, 1: 128.5 us at 1M filters)
#endif
// SNCP: the instantiation that also handles SNC (PreparePNT) and Predict(); the plain update stays free of their
// branches and registers (it is the D(ii) benchmark path).
// PAD: the batch's run-time dimensions (a.n <= NS, a.p <= NM) on zero-padded operands, an identity block in R (kb_vanilla_reg.h PAD):
// every real entry of the result is unchanged, only loads, stores and the condition test see the real sizes.
template <typename T, int NS, int NM, bool EKF, bool FULL, bool EXT, bool SNCP = false, bool PAD = false>
__global__ void __launch_bounds__(64 * HYB_WPB, ((SNCP && FULL) || (PAD && NS > 4)) ? 1 : 2) hybrid_reg_kernel(const StepArgs a) {
    constexpr int TR = tri(NS);
    const int rn = PAD ? a.n : NS, rp = PAD ? a.p : NM;
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * HYB_WPB + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const bool active = tile * KB_TILE + lane < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (rn + tri(rn))) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *yr = (const T *)a.y + tile * a.y_ts + lane;
    const T *yc = (const T *)a.y2 + tile * a.y2_ts + lane;
    T x[NS], P[TR], F[NS * NS];
    const int64_t fi = tile * KB_TILE + lane;
    const T *ephi = EXT ? (const T *)a.ext_phi + (active ? fi : 0) : nullptr;
    const T *eh = EXT ? (const T *)a.ext_h + (active ? fi : 0) : nullptr;
    // Request order "slowest first" (kb_vanilla_reg.h): Phi, Htilde, R and the observations are HBM streams, x and P are
    // Infinity-Cache hits and go last.  (The SNC / Predict instantiation reads Htilde, R and the observations where it needs them.)
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++)
            F[i * NS + j] = (i < rn && j < rn) ? (EXT ? __builtin_nontemporal_load(ephi + (int64_t)(i * rn + j) * a.ext_ld) : ldnt(mo, a.L.mo_F + i * rn + j)) : T(0);
    T H[NM * NS], R[tri(NM)], real[NM], yv[NM];
    if constexpr (!SNCP) {
#pragma unroll
        for (int r = 0; r < NM; r++)
#pragma unroll
            for (int l = 0; l < NS; l++)
                H[r * NS + l] = (r < rp && l < rn) ? (EXT ? __builtin_nontemporal_load(eh + (int64_t)(r * rn + l) * a.ext_ld) : ldnt(mo, a.L.mo_H + r * rn + l)) : T(0);
#pragma unroll
        for (int c = 0; c < NM; c++)
#pragma unroll
            for (int r = 0; r <= c; r++) R[symi(r, c)] = (c < rp) ? ldnt(mo, a.L.mo_R + symi(r, c)) : (r == c ? T(1) : T(0));
#pragma unroll
        for (int r = 0; r < NM; r++) {
            real[r] = (active && r < rp) ? __builtin_nontemporal_load(yr + (int64_t)r * a.y_es) : T(0);
            const T cv = (active && r < rp) ? __builtin_nontemporal_load(yc + (int64_t)r * a.y2_es) : T(0);
            yv[r] = real[r] - cv;
        }
    }
    auto load_state = [&](auto NT) {   // cache policy of the state block: kb_vanilla_reg.h
        constexpr bool nt = decltype(NT)::value;
#pragma unroll
        for (int i = 0; i < NS; i++) x[i] = (i < rn) ? ldp<nt>(st, i) : T(0);
#pragma unroll
        for (int j = 0; j < NS; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) P[symi(i, j)] = (j < rn) ? ldp<nt>(st, rn + symi(i, j)) : T(0);   // (the packed index does not depend on n)
    };
    KB_WITH_STATE_POLICY(a, load_state);
    __builtin_amdgcn_sched_barrier(0);
    // (the arithmetic of the step -- PBar .. the non-finite screen -- is kb_hybrid_reg_step.inc: this kernel and the time-fused one of
    // kb_hybrid_fused.hip compile the SAME text)
#include "kb_hybrid_reg_step.inc"
    if (active && !err) {
        auto store_state = [&](auto NT) {
            constexpr bool nt = decltype(NT)::value;
#pragma unroll
            for (int i = 0; i < NS; i++)
                if (i < rn) stp<nt>(st, i, xn[i]);
#pragma unroll
            for (int j = 0; j < NS; j++)
#pragma unroll
                for (int i = 0; i <= j; i++)
                    if (j < rn) stp<nt>(st, rn + symi(i, j), Pn[symi(i, j)]);
        };
        KB_WITH_STATE_POLICY(a, store_state);
        if constexpr (FULL) {
            T *es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane;
#pragma unroll
            for (int j = 0; j < NS; j++)
#pragma unroll
                for (int i = 0; i <= j; i++)
                    if (j < rn) stnt(es, a.L.es_ppred + symi(i, j), Pm[symi(i, j)]);
#pragma unroll
            for (int i = 0; i < NS; i++)
#pragma unroll
                for (int c = 0; c < NM; c++)
                    if (i < rn && c < rp) stnt(es, a.L.es_gain + i * a.pmax + c, K[i * NM + c]);
#pragma unroll
            for (int r = 0; r < NM; r++)
                if (r < rp) { stnt(es, a.L.es_innov + r, innov[r]); stnt(es, a.L.es_yhat + r, real[r]); stnt(es, a.L.es_dobs + r, yv[r]); }
        }
    }
    if (active && err) fail_step(a, tile * KB_TILE + lane, err);   // hybrid.go:150-152 returns before kf.step++
}

static inline bool hybrid_shape_ok(const StepArgs &a, int NS, int NM, bool pad = false) {
    if (a.flags & KB_FLAG_STATEMENT_KERNELS) return false;
    return (pad ? (a.n <= NS && a.p <= NM) : (a.n == NS && a.p == NM)) && (!a.snc || a.L.nq <= 3) && !(a.flags & KB_FLAG_STRICT_SYMCHECK);
}

template <typename T, int NS, int NM, bool PAD = false>
static inline bool hybrid_try(const Batch &b, const StepArgs &a) {
    if (!hybrid_shape_ok(a, NS, NM, PAD)) return false;
    const dim3 grid((unsigned)((a.ntiles + HYB_WPB - 1) / HYB_WPB)), block(64 * HYB_WPB);
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    const bool sncp = a.snc || a.predict;
#define KB_H(E_, F_) do { if (sncp) { if (a.ext_phi) KB_LAUNCH((hybrid_reg_kernel<T, NS, NM, E_, F_, true, true, PAD>), grid, block, 0, b.stream, a); \
                                      else KB_LAUNCH((hybrid_reg_kernel<T, NS, NM, E_, F_, false, true, PAD>), grid, block, 0, b.stream, a); } \
                          else if (a.ext_phi) KB_LAUNCH((hybrid_reg_kernel<T, NS, NM, E_, F_, true, false, PAD>), grid, block, 0, b.stream, a); \
                          else KB_LAUNCH((hybrid_reg_kernel<T, NS, NM, E_, F_, false, false, PAD>), grid, block, 0, b.stream, a); } while (0)
    if (a.ekf) { if (full) KB_H(true, true); else KB_H(true, false); }
    else       { if (full) KB_H(false, true); else KB_H(false, false); }
#undef KB_H
    return true;
}

}  // namespace kb
