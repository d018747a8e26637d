// kb_information_reg.hip -- register-resident Information filter step (information.go:153-227)
// for the benchmark shape (n = 6, p = 3, fp64) and the padded family up to 6 / 4 / 2.  FULL (KB_FLAG_FULL_ESTIMATE) also
// writes I- and yhat = H State(prev) [+ Measurement(k) for an AWGN batch, information.go:192-194], which costs one more
// n x n inverse per step (State() inverts I on every call, information.go:257-293); the stale-1x1-Rinv quirk is a template flag.
// Per filter-step it reads i[n], I (packed), F^-1 [n^2], Q^-1 and R^-1 (packed: the upper triangle of the computed
// inverse, mirrored), H [p n], y[p] and writes i, I.  The (M + Q^-1)^-1 inverse is the LU-pivoted register inverse (kb_device.h).
#include "kb_internal.h"
#include "kb_static.h"
#include "kb_vanilla_reg.h"   // draw_normals / chol_times / TilePtr (the AWGN draw of the register kernels)

namespace kb {
#ifndef INFO_WPB
#define INFO_WPB 1   // waves per workgroup
#endif
#ifndef INFO_WAVES
#define INFO_WAVES 2
#endif
#ifndef INFO_STASH
#define INFO_STASH 1
#endif
template <typename T, int NS>
constexpr bool info_stash() { return INFO_STASH && sizeof(T) == 8 && NS > 4; }


// PAD: run-time dimensions a.n <= NS, a.p <= NM, a.m <= NC on operands padded with zeros and an identity block in
// Q^-1 (so that M + Q^-1 stays invertible; Z, I- and i- keep exact zeros in the padding): only loads and stores see
// the real sizes (cf. kb_vanilla_reg.h).
// SHARED: one model for all filters (StepArgs::mo_ts == 0), the model operands come from lane 0's copy in tile 0's block with the
// default cache policy (wave-uniform addresses: scalar loads where no store precedes them; kb_vanilla_reg.h ldm)
template <typename T, int NS, int NM, int NC, bool SCALAR_RINV, bool PAD = false, bool FULL = false, bool NOISE = false, bool SHARED = false>
__global__ void __launch_bounds__(64 * INFO_WPB, INFO_WAVES) information_reg_kernel(const StepArgs a) {
    static_assert(FULL || !NOISE, "Information draws Measurement(k) only: nothing but the FULL estimate's yhat sees it");
    constexpr int TR = tri(NS);
    constexpr bool STASH = info_stash<T, NS>();
    const int rn = PAD ? a.n : NS, rp = PAD ? a.p : NM, rm = PAD ? a.m : NC;
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * INFO_WPB + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const bool active = tile * KB_TILE + lane < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (rn + tri(rn))) + lane;
    // one model for all filters (StepArgs::mo_ts == 0): every lane reads lane 0's copy in tile 0's block -- 8 bytes per load, not a 512-byte row
    const T *mo = SHARED ? (const T *)a.model : (const T *)a.model + tile * a.mo_ts + (a.mo_ts ? lane : 0);
    auto ldmo = [&](const T *q, int e) __attribute__((always_inline)) { return SHARED ? q[(int64_t)e * KB_TILE] : ldnt(q, e); };
    const T *yp = (const T *)a.y + tile * a.y_ts + lane;
    // request order "slowest first" (kb_vanilla_reg.h): F^-1 is an HBM stream, i and I are Infinity-Cache hits
    T iv[NS], I[TR], Fi[NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) Fi[i * NS + j] = (i < rn && j < rn) ? ldmo(mo, a.L.mo_Finv + i * rn + j) : T(0);
    auto load_state = [&](auto NT) {   // cache policy of the state block: kb_vanilla_reg.h
        constexpr bool nt = decltype(NT)::value;
#pragma unroll
        for (int i = 0; i < NS; i++) iv[i] = (i < rn) ? ldp<nt>(st, i) : T(0);
#pragma unroll
        for (int j = 0; j < NS; j++)
#pragma unroll
            for (int i = 0; i <= j; i++) I[symi(i, j)] = (j < rn) ? ldp<nt>(st, rn + symi(i, j)) : T(0);
    };
    KB_WITH_STATE_POLICY(a, load_state);
    __builtin_amdgcn_sched_barrier(0);
    // :163-165 zk = Finv^T (I Finv), one column at a time: column j of I Finv lives only until column j of zk is formed
    // (the whole intermediate product would put 3 n^2 + n(n+1)/2 doubles in registers at once)
    T zk[NS * NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        T t1[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += I[symi(i, l)] * Fi[l * NS + j];
            t1[i] = s;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += Fi[l * NS + i] * t1[l];
            zk[i * NS + j] = s;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) pin(zk[i * NS + j]);
    }
    // iKp1Minus (first half, :176-177): Finv^T i
    T im[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        T s = T(0);
#pragma unroll
        for (int i = 0; i < NS; i++) s += Fi[i * NS + j] * iv[i];
        im[j] = s;
    }
    // pin (kb_device.h): without it the products above are sunk into the basic blocks of the pivoted solve below and
    // I, Finv, t1 stay alive beside its two work arrays
#pragma unroll
    for (int i = 0; i < NS * NS; i++) pin(zk[i]);
#pragma unroll
    for (int i = 0; i < NS; i++) pin(im[i]);
    __builtin_amdgcn_sched_barrier(0);
    // :169-174 Z = -zk (zk + Qinv)^-1 (inverse error ignored by the reference).  Obtained as an in-place
    // pivoted LU solve of (zk + Qinv)^T X = zk^T (X = (zk (zk + Qinv)^-1)^T) instead of inverse-then-multiply:
    // two 6x6 work arrays instead of four (the explicit inverse kept the kernel at 1 wave/SIMD).
    T zqT[NS * NS], X[NS * NS], Z[NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            zqT[j * NS + i] = zk[i * NS + j] + ((i < rn && j < rn) ? ldmo(mo, a.L.mo_Qinv + symi(i, j)) : (i == j ? T(1) : T(0)));
            X[j * NS + i] = zk[i * NS + j];
        }
#pragma unroll
    for (int i = 0; i < NS * NS; i++) pin(zqT[i]);
    // zk is needed again for I- (:188-190) but not by the solve: with the solve's two work arrays it would make 3 n^2
    // live doubles (228 registers at n = 6).  It waits in LDS instead (n^2 x 512 B per wave, 8 waves per CU = 147 KB).
    __shared__ T lds[STASH ? NS * NS * 64 * INFO_WPB : 1];
    [[maybe_unused]] const int so = (threadIdx.x >> 6) * (NS * NS * 64) + lane;
    if constexpr (STASH) {
#pragma unroll
        for (int i = 0; i < NS * NS; i++) lds[so + i * 64] = zk[i];
        asm volatile("" ::: "memory");  // no store-to-load forwarding: the point is to free the registers
    }
    __builtin_amdgcn_sched_barrier(0);
    lu_solve_inplace<T, NS, NS>(zqT, X);
#pragma unroll
    for (int i = 0; i < NS * NS; i++) pin(X[i]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STASH) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < NS * NS; i++) zk[i] = lds[so + i * 64];
    }
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) Z[i * NS + j] = T(-1) * X[j * NS + i];
    if constexpr (NC > 0) {  // :178-182 i- += zk (G u)
        const T *up = (const T *)a.u + tile * a.u_ts + lane;
        T gu[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int c = 0; c < NC; c++)
                if (i < rn && c < rm) s += ldmo(mo, a.L.mo_G + i * rm + c) * (active ? __builtin_nontemporal_load(up + (int64_t)c * a.u_es) : T(0));
            gu[i] = s;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NS; j++) s += zk[i * NS + j] * gu[j];
            im[i] = im[i] + s;
        }
    }
    // :183-185 i- = (1 + Z) i-
    T imn[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < NS; j++) s += ((i == j ? T(1) : T(0)) + Z[i * NS + j]) * im[j];
        imn[i] = s;
    }
    // :188-190 I- = zk + Z zk^T (upper triangle)
    T Im[TR];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = i; j < NS; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += Z[i * NS + l] * zk[j * NS + l];
            Im[symi(i, j)] = zk[i * NS + j] + s;
        }
    if constexpr (FULL) {   // I- leaves at once (Estimate.PredCovariance, information.go:295-316 inverts it lazily)
        T *es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane;
        if (active) {
#pragma unroll
            for (int i = 0; i < NS; i++)
#pragma unroll
                for (int j = i; j < NS; j++)
                    if (j < rn) stnt(es, a.L.es_ppred + symi(i, j), Im[symi(i, j)]);
        }
    }
    // H and R^-1 are only requested once zk and Z are dead (the other wave of the SIMD covers the latency)
#pragma unroll
    for (int i = 0; i < TR; i++) pin(Im[i]);
#pragma unroll
    for (int i = 0; i < NS; i++) pin(imn[i]);
    __builtin_amdgcn_sched_barrier(0);
    // :197-212 HTR = H^T Rinv; i+ = HTR y + i-; I+ = I- + HTR H
    T H[NM * NS], HTR[NS * NM];
#pragma unroll
    for (int r = 0; r < NM; r++)
#pragma unroll
        for (int l = 0; l < NS; l++) H[r * NS + l] = (r < rp && l < rn) ? ldmo(mo, a.L.mo_H + r * rn + l) : T(0);
    if constexpr (SCALAR_RINV) {  // QUIRK information.go:198-200: a (stale) 1x1 R^-1 scales H^T whatever p is
        const T r0 = ldmo(mo, a.L.mo_Rinv);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NM; j++) HTR[i * NM + j] = r0 * H[j * NS + i];
    } else {
        T Ri[NM * NM];
#pragma unroll
        for (int l = 0; l < NM; l++)
#pragma unroll
            for (int j = 0; j < NM; j++) Ri[l * NM + j] = (l < rp && j < rp) ? ldmo(mo, a.L.mo_Rinv + symi(l, j)) : T(0);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NM; j++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NM; l++) s += H[l * NS + i] * Ri[l * NM + j];
                HTR[i * NM + j] = s;
            }
    }
    T chk = T(0);
    T ip[NS], Ip[TR];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < NM; j++) {
            const T yv = (active && j < rp) ? __builtin_nontemporal_load(yp + (int64_t)j * a.y_es) : T(0);
            s += HTR[i * NM + j] * yv;
        }
        ip[i] = s + imn[i];
        chk += ip[i] * T(0);
#pragma unroll
        for (int j = i; j < NS; j++) {
            T s2 = T(0);
#pragma unroll
            for (int l = 0; l < NM; l++) s2 += HTR[i * NM + l] * H[l * NS + j];
            Ip[symi(i, j)] = Im[symi(i, j)] + s2;
            chk += Ip[symi(i, j)] * T(0);
        }
    }
    const bool ok = !(chk != chk);
    [[maybe_unused]] T iprev[NS], Iprev[NS * NS];
    if constexpr (FULL) {
        // information.go:192-194 yhat = H State(prev) + Measurement(k), State() = inverse(I) i with the inverse mirrored from
        // its upper triangle (AsSymDense), or ZEROS when gonum's Inverse reports a Condition error (singular, or cond > 1e16:
        // information.go:284-288 -- the zero-covariance-until-observable rows of examples/jerkcar/information.csv).  The previous
        // (i, I) are read again here (cache hits), BEFORE the state block is rewritten; H is requested again afterwards.
#pragma unroll
        for (int i = 0; i < NS; i++) pin(ip[i]);
#pragma unroll
        for (int e = 0; e < TR; e++) pin(Ip[e]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NS; i++) iprev[i] = (i < rn) ? ldt(st, i) : T(0);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) Iprev[i * NS + j] = (i < rn && j < rn) ? ldt(st, rn + symi(i, j)) : (i == j ? T(1) : T(0));
#pragma unroll
        for (int i = 0; i < NS; i++) pin(iprev[i]);
#pragma unroll
        for (int e = 0; e < NS * NS; e++) pin(Iprev[e]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (active && ok) {
        auto store_state = [&](auto NT) {
            constexpr bool nt = decltype(NT)::value;
#pragma unroll
            for (int i = 0; i < NS; i++)
                if (i < rn) stp<nt>(st, i, ip[i]);
#pragma unroll
            for (int j = 0; j < NS; j++)
#pragma unroll
                for (int i = 0; i <= j; i++)
                    if (j < rn) stp<nt>(st, rn + symi(i, j), Ip[symi(i, j)]);
        };
        KB_WITH_STATE_POLICY(a, store_state);
    }
    if (active && !ok) atomicOr(a.status + tile * KB_TILE + lane, (unsigned)KB_ST_NONFINITE);
    if constexpr (FULL) {
        __builtin_amdgcn_sched_barrier(0);
        T Pp[NS * NS], xp[NS], yhat[NM];
        // H (and chol R) are read again here, long after the top of the kernel: through anchored() (kb_device.h), or their 18 + 6
        // vector addresses are formed up there, carried through the 6 x 6 inverse and spilled (180 B of scratch, one wait per reload)
        const int64_t tile_u = (int64_t)blockIdx.x * INFO_WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // the tile index as a scalar
        const T *const mo_u = SHARED ? (const T *)a.model : (const T *)a.model + tile_u * a.mo_ts;
        if constexpr (STASH) {   // the previous i sits out the inverse where zk sat out the solve
#pragma unroll
            for (int i = 0; i < NS; i++) lds[so + i * 64] = iprev[i];
            asm volatile("" ::: "memory");
        }
        const bool bad = inverse_lu<T, NS>(Iprev, Pp, rn);
#pragma unroll
        for (int e = 0; e < NS * NS; e++) pin(Pp[e]);
        if constexpr (STASH) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < NS; i++) iprev[i] = lds[so + i * 64];
        }   // (the lane index below is a volatile asm: only other asm statements hold it behind the inverse)
        __builtin_amdgcn_sched_barrier(0);
        const unsigned mo_lane = (SHARED || a.mo_ts == 0) ? 0u : (late_lane() & 63u);   // (formed behind the inverse, not carried through it)
        auto late_mo = [&](int rt, int c) __attribute__((always_inline)) {
            const auto gp = anchored(mo_u, rt, c) + mo_lane;
            return SHARED ? *gp : __builtin_nontemporal_load(gp);
        };
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NS; j++) s += (bad ? T(0) : Pp[(i <= j ? i : j) * NS + (i <= j ? j : i)]) * iprev[j];
            xp[i] = s;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) pin(xp[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < NM; r++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += ((r < rp && l < rn) ? late_mo(a.L.mo_H + (PAD ? r * rn : 0), (PAD ? 0 : r * NS) + l) : T(0)) * xp[l];
            yhat[r] = s;
        }
        if constexpr (NOISE) {
#pragma unroll
            for (int r = 0; r < NM; r++) pin(yhat[r]);
            __builtin_amdgcn_sched_barrier(0);
            const int64_t fi_u = tile_u * KB_TILE + (late_lane() & 63u);   // (the filter index formed again: carried from the top it is a spilled pair)
            const uint64_t gfi = (uint64_t)(a.first_filter + fi_u);
            const uint32_t stepno = (uint32_t)a.step0 - (active ? a.lag[fi_u] : 0u);   // kf.step of this filter
            T z1[NM], v[NM];
            draw_normals<T, NM>(a, gfi, stepno, 1u, z1);
#pragma unroll
            for (int i = 0; i < NM; i++) {   // v = chol(R) z (chol_times, kb_vanilla_reg.h), the factor read through anchored()
                T sacc = T(0);
#pragma unroll
                for (int kk = 0; kk <= i; kk++) sacc += ((i < rp) ? late_mo(a.L.mo_LR, symi(kk, i)) : T(0)) * z1[kk];
                v[i] = sacc;
            }
#pragma unroll
            for (int r = 0; r < NM; r++) yhat[r] += v[r];
        }
        T *es = (T *)a.est + tile_u * ((int64_t)KB_TILE * a.L.es_elems) + (late_lane() & 63u);   // (formed here: kept from the top it costs a spilled register pair)
        if (active && ok) {
#pragma unroll
            for (int r = 0; r < NM; r++)
                if (r < rp) stnt(es, a.L.es_yhat + r, yhat[r]);
        }
    }
}

// Which instantiation a batch needs: the draws of an Information step only reach yhat (information.go:194), so a batch
// without KB_FLAG_FULL_ESTIMATE runs the plain kernel whatever its Noise is; BatchNoise stays on the generic kernel.
static bool info_variant(const StepArgs &a, bool &full, bool &noise) {
    if (a.flags & KB_FLAG_STRICT_SYMCHECK) return false;
    full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    noise = full && a.noise_kind == KB_NOISE_AWGN;
    return a.noise_kind != KB_NOISE_BATCH;
}
#define KB_INFO_GO(SC_, PAD_)                                                                                                        \
    do {                                                                                                                             \
        if (!full) KB_LAUNCH((information_reg_kernel<T, NS, NM, NC, SC_, PAD_, false, false, SHARED>), grid, block, 0, b.stream, a); \
        else if (!noise) KB_LAUNCH((information_reg_kernel<T, NS, NM, NC, SC_, PAD_, true, false, SHARED>), grid, block, 0, b.stream, a); \
        else KB_LAUNCH((information_reg_kernel<T, NS, NM, NC, SC_, PAD_, true, true, SHARED>), grid, block, 0, b.stream, a);      \
    } while (0)

template <typename T, int NS, int NM, int NC = 0, bool SHARED = false>
static bool info_try(const Batch &b, const StepArgs &a) {
    bool full = false, noise = false;
    if (SHARED && (a.mo_ts != 0 || (a.flags & KB_FLAG_FULL_ESTIMATE))) return false;   // (the FULL variants stay on the general instantiations)
    if (a.n != NS || a.p != NM || (a.rinv_p != NM && a.rinv_p != 1) || (a.need_ctrl ? a.m : 0) != NC || a.nsteps != 1 || !info_variant(a, full, noise))
        return false;
    const dim3 grid((unsigned)((a.ntiles + INFO_WPB - 1) / INFO_WPB)), block(64 * INFO_WPB);
    if (a.rinv_p == 1) KB_INFO_GO(true, false);
    else KB_INFO_GO(false, false);
    return true;
}

// any (n, p, m) with n <= NS, p <= NM, m <= NC (NC == 0 iff no control input) on the padded instantiation
template <typename T, int NS, int NM, int NC, bool SHARED = false>
static bool info_try_pad(const Batch &b, const StepArgs &a) {
    const int m = a.need_ctrl ? a.m : 0;
    bool full = false, noise = false;
    if (SHARED && (a.mo_ts != 0 || (a.flags & KB_FLAG_FULL_ESTIMATE))) return false;
    if (a.n > NS || a.p > NM || (a.rinv_p != a.p && a.rinv_p != 1) || m > NC || (NC == 0) != (m == 0) || a.nsteps != 1 || !info_variant(a, full, noise))
        return false;
    const dim3 grid((unsigned)((a.ntiles + INFO_WPB - 1) / INFO_WPB)), block(64 * INFO_WPB);
    if (a.rinv_p == 1 && a.p != 1) KB_INFO_GO(true, true);
    else KB_INFO_GO(false, true);
    return true;
}

int launch_information(const Batch &b, const StepArgs &a) {
    if (a.flags & KB_FLAG_STATEMENT_KERNELS) return launch_information_gen(b, a);
    bool done = false;
    if (b.dtype == KB_F64 && a.mo_ts == 0)   // one model for all filters: the SHARED instantiations (state-only outputs)
        done = info_try<double, 6, 3, 0, true>(b, a) || info_try<double, 4, 2, 0, true>(b, a) || info_try_pad<double, 4, 2, 0, true>(b, a) ||
               info_try_pad<double, 4, 2, 2, true>(b, a) || info_try_pad<double, 6, 4, 0, true>(b, a) || info_try_pad<double, 6, 4, 2, true>(b, a);
    if (!done && b.dtype == KB_F64)
        done = info_try<double, 6, 3>(b, a) || info_try<double, 4, 2>(b, a) ||
               info_try<double, 4, 1, 1>(b, a) || info_try<double, 4, 2, 1>(b, a);  // examples/jerkcar
    if (!done && b.dtype == KB_F64)   // shapes without an exact instantiation: padded register kernels up to 6 / 4 / 2
        done = info_try_pad<double, 4, 2, 0>(b, a) || info_try_pad<double, 4, 2, 2>(b, a) || info_try_pad<double, 6, 4, 0>(b, a) ||
               info_try_pad<double, 6, 4, 2>(b, a);
    if (!done) done = launch_information_split(b, a);   // 6 < n <= 16 (p <= 8, m <= 2), state-only outputs: kb_information_split.h
    if (!done) return launch_information_gen(b, a);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
