// kb_information_reg.hip -- register-resident Information filter step (information.go:153-227)
// for the benchmark shape (n = 6, p = 3, fp64), state-only outputs (i+, I+).  Batches created
// with KB_FLAG_FULL_ESTIMATE (which also need yhat = H State(prev), i.e. one more n x n inverse,
// and I-) and the stale-1x1-Rinv quirk go through the generic kernel.
// Per filter-step it reads i[n], I (packed), F^-1 [n^2], Q^-1 [n^2], H [p n], R^-1 [p^2], y[p] and
// writes i, I.  The (M + Q^-1)^-1 inverse is the LU-pivoted register inverse (kb_device.h).
#include "kb_internal.h"
#include "kb_static.h"

namespace kb {
#ifndef INFO_WPB
#define INFO_WPB 1   // waves per workgroup
#endif
#ifndef INFO_WAVES
#define INFO_WAVES 1
#endif


template <typename T, int NS, int NM, int NC, bool SCALAR_RINV>
__global__ void __launch_bounds__(64 * INFO_WPB, INFO_WAVES) information_reg_kernel(const StepArgs a) {
    constexpr int TR = tri(NS);
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * INFO_WPB + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const bool active = tile * KB_TILE + lane < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (NS + TR)) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *yp = (const T *)a.y + tile * a.y_ts + lane;
    T iv[NS], I[TR], Fi[NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++) iv[i] = ldt(st, i);
#pragma unroll
    for (int e = 0; e < TR; e++) I[e] = ldt(st, NS + e);
#pragma unroll
    for (int e = 0; e < NS * NS; e++) Fi[e] = ldnt(mo, a.L.mo_Finv + e);
    // :163-165 zk = Finv^T (I Finv)
    T t1[NS * NS], zk[NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += I[symi(i, l)] * Fi[l * NS + j];
            t1[i * NS + j] = s;
        }
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += Fi[l * NS + i] * t1[l * NS + j];
            zk[i * NS + j] = s;
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
    // :169-174 Z = -zk (zk + Qinv)^-1 (inverse error ignored by the reference).  Obtained as an in-place
    // pivoted LU solve of (zk + Qinv)^T X = zk^T (X = (zk (zk + Qinv)^-1)^T) instead of inverse-then-multiply:
    // two 6x6 work arrays instead of four (the explicit inverse kept the kernel at 1 wave/SIMD).
    T zqT[NS * NS], X[NS * NS], Z[NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            zqT[j * NS + i] = zk[i * NS + j] + ldnt(mo, a.L.mo_Qinv + i * NS + j);
            X[j * NS + i] = zk[i * NS + j];
        }
    lu_solve_inplace<T, NS, NS>(zqT, X);
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
            for (int c = 0; c < NC; c++) s += ldnt(mo, a.L.mo_G + i * NC + c) * (active ? __builtin_nontemporal_load(up + (int64_t)c * a.u_es) : T(0));
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
    // :197-212 HTR = H^T Rinv; i+ = HTR y + i-; I+ = I- + HTR H
    T H[NM * NS], HTR[NS * NM];
#pragma unroll
    for (int e = 0; e < NM * NS; e++) H[e] = ldnt(mo, a.L.mo_H + e);
    if constexpr (SCALAR_RINV) {  // QUIRK information.go:198-200: a (stale) 1x1 R^-1 scales H^T whatever p is
        const T r0 = ldnt(mo, a.L.mo_Rinv);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NM; j++) HTR[i * NM + j] = r0 * H[j * NS + i];
    } else {
        T Ri[NM * NM];
#pragma unroll
        for (int e = 0; e < NM * NM; e++) Ri[e] = ldnt(mo, a.L.mo_Rinv + e);
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
            const T yv = active ? __builtin_nontemporal_load(yp + (int64_t)j * a.y_es) : T(0);
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
    if (active && ok) {
#pragma unroll
        for (int i = 0; i < NS; i++) stt(st, i, ip[i]);
#pragma unroll
        for (int e = 0; e < TR; e++) stt(st, NS + e, Ip[e]);
    }
    if (active && !ok) atomicOr(a.status + tile * KB_TILE + lane, (unsigned)KB_ST_NONFINITE);
}

template <typename T, int NS, int NM, int NC = 0>
static bool info_try(const Batch &b, const StepArgs &a) {
    if (a.n != NS || a.p != NM || (a.rinv_p != NM && a.rinv_p != 1) || (a.need_ctrl ? a.m : 0) != NC || a.nsteps != 1 ||
        (a.flags & (KB_FLAG_FULL_ESTIMATE | KB_FLAG_STRICT_SYMCHECK)) || a.noise_kind != KB_NOISE_NOISELESS)
        return false;
    if (a.rinv_p == 1) hipLaunchKernelGGL((information_reg_kernel<T, NS, NM, NC, true>), dim3((unsigned)((a.ntiles + INFO_WPB - 1) / INFO_WPB)), dim3(64 * INFO_WPB), 0, b.stream, a);
    else hipLaunchKernelGGL((information_reg_kernel<T, NS, NM, NC, false>), dim3((unsigned)((a.ntiles + INFO_WPB - 1) / INFO_WPB)), dim3(64 * INFO_WPB), 0, b.stream, a);
    return true;
}

int launch_information(const Batch &b, const StepArgs &a) {
    bool done = false;
    if (b.dtype == KB_F64)
        done = info_try<double, 6, 3>(b, a) || info_try<double, 4, 2>(b, a) ||
               info_try<double, 4, 1, 1>(b, a) || info_try<double, 4, 2, 1>(b, a);  // examples/jerkcar
    if (!done) return launch_information_gen(b, a);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
