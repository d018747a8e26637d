// kb_chisq.hip -- NEES / NIS consistency statistics (chisquare.go:16-95) fused with the
// Monte-Carlo truth generation (montecarlo.go:92-119).
//
// Reference flow: NewMonteCarloRuns produces `runs` truth trajectories with a pure-predictor
// Vanilla + AWGN (state x_k and measurement yhat_k = H x_{k-1} + v_k per step); NewChiSquare then
// Reset()s a full filter per run, replays `kf.Update(truth.Measurement(), u_k)` and accumulates
//   NEES_k = (x_k - xhat_k)^T P_k^-1 (x_k - xhat_k)          (chisquare.go:46-59)
//   NIS_k  = innov^T (H P-_k H^T + R)^-1 innov               (chisquare.go:61-77)
// averaged over runs per step.  Here one lane = one run: the truth and the filter advance
// together in registers, all steps inside one launch; per step the wave reduces both statistics
// (__shfl_xor) and adds them to one of 32 replicas with fp64 atomics.  The truth's noise comes
// from the same Philox stream as kb_mc_run (seed; global run index, step, epoch, draw), so the
// statistics refer to the same runs as the Monte-Carlo means when the same epoch is replayed.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kb_internal.h"
#include "kb_static.h"

namespace kb {

constexpr int CHI_REPL = 32;



struct ChiArgs {
    const void *t_state, *t_model;  // truth batch (pure predictor, AWGN factors in its model block)
    const void *k_state, *k_model;  // filter batch (initial estimate + model)
    Layout tL, kL;
    int64_t N, ntiles, first_run, epoch;
    uint64_t seed;
    int nsteps, ncontrols, need_ctrl, with_nees, with_nis;
    const void *controls;
    double *sums;  // [CHI_REPL][steps][2]
};

template <typename T, int NS, int NM, int NC>
__global__ void __launch_bounds__(256, (NS <= 4 ? 2 : 1)) chisq_kernel(const ChiArgs a) {
    constexpr int TR = tri(NS), TM = tri(NM);
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    const T *ts = (const T *)a.t_state + tile * ((int64_t)KB_TILE * a.tL.st_elems) + lane;
    const T *tm = (const T *)a.t_model + tile * ((int64_t)KB_TILE * a.tL.mo_elems) + lane;
    const T *ks = (const T *)a.k_state + tile * ((int64_t)KB_TILE * a.kL.st_elems) + lane;
    const T *km = (const T *)a.k_model + tile * ((int64_t)KB_TILE * a.kL.mo_elems) + lane;
    // truth model
    T xt[NS], Ft[NS * NS];
    // the truth model's H, chol(Q), chol(R) are read once per step: they wait in LDS ((NM NS + TR + TM) x 2 KB per workgroup)
    // with a control input the two G (read once per step as well) wait there too: registers are what this kernel is short of
    constexpr int NPARK = NM * NS + TR + TM + 2 * NS * NC;
    __shared__ T park[NPARK * 256];
    T *pk = park + threadIdx.x;
#define Ht(e) pk[(e) * 256]
#define LQ(e) pk[(NM * NS + (e)) * 256]
#define LR(e) pk[(NM * NS + TR + (e)) * 256]
#define Gt(e) pk[(NM * NS + TR + TM + (e)) * 256]
#define Gk(e) pk[(NM * NS + TR + TM + NS * NC + (e)) * 256]
#pragma unroll
    for (int i = 0; i < NS; i++) xt[i] = ldt(ts, a.tL.st_vec + i);
#pragma unroll
    for (int e = 0; e < NS * NS; e++) Ft[e] = ldt(tm, a.tL.mo_F + e);
#pragma unroll
    for (int e = 0; e < NM * NS; e++) Ht(e) = ldt(tm, a.tL.mo_H + e);
#pragma unroll
    for (int e = 0; e < TR; e++) LQ(e) = ldt(tm, a.tL.mo_LQ + e);
#pragma unroll
    for (int e = 0; e < TM; e++) LR(e) = ldt(tm, a.tL.mo_LR + e);
    // filter
    T x[NS], P[TR], F[NS * NS], H[NM * NS], Q[TR], R[TM];
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = ldt(ks, a.kL.st_vec + i);
#pragma unroll
    for (int e = 0; e < TR; e++) P[e] = ldt(ks, a.kL.st_mat + e);
#pragma unroll
    for (int e = 0; e < NS * NS; e++) F[e] = ldt(km, a.kL.mo_F + e);
#pragma unroll
    for (int e = 0; e < NM * NS; e++) H[e] = ldt(km, a.kL.mo_H + e);
#pragma unroll
    for (int e = 0; e < TR; e++) Q[e] = ldt(km, a.kL.mo_Q + e);
#pragma unroll
    for (int e = 0; e < TM; e++) R[e] = ldt(km, a.kL.mo_R + e);
    if constexpr (NC > 0) {
#pragma unroll
        for (int e = 0; e < NS * NC; e++) { Gt(e) = ldt(tm, a.tL.mo_G + e); Gk(e) = ldt(km, a.kL.mo_G + e); }
    }
    const uint64_t gfi = (uint64_t)(a.first_run + fi);
    double *my = a.sums + (size_t)(tile % CHI_REPL) * a.nsteps * 2;
    for (int t = 0; t < a.nsteps; t++) {
        asm volatile("" ::: "memory");   // the parked operands are re-read each step, not hoisted back into registers
        [[maybe_unused]] T u[NC > 0 ? NC : 1];
        if constexpr (NC > 0) {
            const T *up = (const T *)a.controls + (a.ncontrols == 1 ? 0 : (int64_t)t * NC);
#pragma unroll
            for (int c = 0; c < NC; c++) u[c] = a.ncontrols == 1 ? T(0) : up[c];
        }
        // ---- truth: yhat_k = H x_{k-1} + v_k ; x_k = F x_{k-1} [+ G u_k] + w_k  (vanilla.go:138-157, predictionOnly)
        T zq[NS], zr[NM];
#pragma unroll
        for (int k2 = 0; k2 < NS; k2 += 2) {
            uint32_t r[4];
            Philox::gen(a.seed, gfi, (uint32_t)t, ((uint32_t)(a.epoch * 4 + 0) << 8) | (uint32_t)(k2 >> 1), r);
            double z0, z1;
            box_muller(r, z0, z1);
            zq[k2] = (T)z0;
            if (k2 + 1 < NS) zq[k2 + 1] = (T)z1;
        }
#pragma unroll
        for (int k2 = 0; k2 < NM; k2 += 2) {
            uint32_t r[4];
            Philox::gen(a.seed, gfi, (uint32_t)t, ((uint32_t)(a.epoch * 4 + 1) << 8) | (uint32_t)(k2 >> 1), r);
            double z0, z1;
            box_muller(r, z0, z1);
            zr[k2] = (T)z0;
            if (k2 + 1 < NM) zr[k2 + 1] = (T)z1;
        }
        T y[NM], xtn[NS];
#pragma unroll
        for (int r2 = 0; r2 < NM; r2++) {
            T s = T(0), v = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += Ht(r2 * NS + l) * xt[l];
#pragma unroll
            for (int k2 = 0; k2 <= r2; k2++) v += LR(symi(k2, r2)) * zr[k2];
            y[r2] = s + v;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0), w = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += Ft[i * NS + l] * xt[l];
            if constexpr (NC > 0) {
                T g = T(0);
#pragma unroll
                for (int c = 0; c < NC; c++) g += Gt(i * NC + c) * u[c];
                s = s + g;
            }
#pragma unroll
            for (int k2 = 0; k2 <= i; k2++) w += LQ(symi(k2, i)) * zq[k2];
            xtn[i] = s + w;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) xt[i] = xtn[i];
#pragma unroll
        for (int i = 0; i < NS; i++) pin(xt[i]);
#pragma unroll
        for (int i = 0; i < NM; i++) pin(y[i]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- filter: Vanilla.Update(y, u), Noiseless (vanilla.go:128-220)
        T xm[NS], Pm[TR];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += F[i * NS + l] * x[l];
            if constexpr (NC > 0) {
                T g = T(0);
#pragma unroll
                for (int c = 0; c < NC; c++) g += Gk(i * NC + c) * u[c];
                s = s + g;
            }
            xm[i] = s;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T fp[NS];
#pragma unroll
            for (int k2 = 0; k2 < NS; k2++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += F[i * NS + l] * P[symi(l, k2)];
                fp[k2] = s;
            }
#pragma unroll
            for (int j = i; j < NS; j++) {
                T s = T(0);
#pragma unroll
                for (int k2 = 0; k2 < NS; k2++) s += fp[k2] * F[j * NS + k2];
                Pm[symi(i, j)] = s + Q[symi(i, j)];
            }
        }
#pragma unroll
        for (int i = 0; i < TR; i++) pin(Pm[i]);
#pragma unroll
        for (int i = 0; i < NS; i++) pin(xm[i]);
        __builtin_amdgcn_sched_barrier(0);
        T PHt[NS * NM], S[NM * NM], Si[NM * NM], K[NS * NM];
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int c = 0; c < NM; c++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += Pm[symi(i, l)] * H[c * NS + l];
                PHt[i * NM + c] = s;
            }
#pragma unroll
        for (int r2 = 0; r2 < NM; r2++)
#pragma unroll
            for (int c = 0; c < NM; c++) {
                T s = T(0);
#pragma unroll
                for (int i = 0; i < NS; i++) s += H[r2 * NS + i] * PHt[i * NM + c];
                S[r2 * NM + c] = s + R[symi(r2, c)];
            }
#pragma unroll
        for (int i = 0; i < NS * NM; i++) pin(PHt[i]);
#pragma unroll
        for (int i = 0; i < NM * NM; i++) pin(S[i]);
        __builtin_amdgcn_sched_barrier(0);
        inverse_lu<T, NM>(S, Si);
        smm_nn<T, NS, NM, NM>(PHt, Si, K);
#pragma unroll
        for (int i = 0; i < NS * NM; i++) pin(K[i]);
#pragma unroll
        for (int i = 0; i < NM * NM; i++) pin(Si[i]);
        __builtin_amdgcn_sched_barrier(0);
        T innov[NM];
#pragma unroll
        for (int r2 = 0; r2 < NM; r2++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += H[r2 * NS + l] * xm[l];
            innov[r2] = y[r2] - s;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int c = 0; c < NM; c++) s += K[i * NM + c] * innov[c];
            x[i] = xm[i] + s;
        }
        T A[NS * NS];
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) {
                T s = T(0);
#pragma unroll
                for (int c = 0; c < NM; c++) s += K[i * NM + c] * H[c * NS + j];
                A[i * NS + j] = (i == j ? T(1) : T(0)) - s;
            }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T ap[NS], kr[NM];
#pragma unroll
            for (int k2 = 0; k2 < NS; k2++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += A[i * NS + l] * Pm[symi(l, k2)];
                ap[k2] = s;
            }
#pragma unroll
            for (int c = 0; c < NM; c++) {
                T s = T(0);
#pragma unroll
                for (int k2 = 0; k2 < NM; k2++) s += K[i * NM + k2] * R[symi(k2, c)];
                kr[c] = s;
            }
#pragma unroll
            for (int j = i; j < NS; j++) {
                T s = T(0), s2 = T(0);
#pragma unroll
                for (int k2 = 0; k2 < NS; k2++) s += ap[k2] * A[j * NS + k2];
#pragma unroll
                for (int c = 0; c < NM; c++) s2 += kr[c] * K[j * NM + c];
                P[symi(i, j)] = s + s2;
            }
        }
#pragma unroll
        for (int i = 0; i < TR; i++) pin(P[i]);
#pragma unroll
        for (int i = 0; i < NS; i++) pin(x[i]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- statistics (chisquare.go:46-77)
        double nis = 0.0, nees = 0.0;
        if (a.with_nis) {
            T s = T(0);
#pragma unroll
            for (int r2 = 0; r2 < NM; r2++) {
                T v = T(0);
#pragma unroll
                for (int c = 0; c < NM; c++) v += Si[r2 * NM + c] * innov[c];
                s += innov[r2] * v;
            }
            nis = (double)s;
        }
        if (a.with_nees) {
            T Pf[NS * NS], Pi[NS * NS], dlt[NS];
#pragma unroll
            for (int i = 0; i < NS; i++) {
                dlt[i] = xt[i] - x[i];
#pragma unroll
                for (int j = 0; j < NS; j++) Pf[i * NS + j] = P[symi(i, j)];
            }
            inverse_lu<T, NS>(Pf, Pi);
            T s = T(0);
#pragma unroll
            for (int i = 0; i < NS; i++) {
                T v = T(0);
#pragma unroll
                for (int j = 0; j < NS; j++) v += Pi[i * NS + j] * dlt[j];
                s += dlt[i] * v;
            }
            nees = (double)s;
        }
        // both sums in one butterfly: lanes swap one of the two values first, so 1 + 5 exchanges instead of 2 x 6; even
        // lanes end with the NIS total, odd lanes with the NEES total, and lanes 0 / 1 add them with one atomic instruction
        const double v0 = active ? nis : 0.0, v1 = active ? nees : 0.0;
        const bool odd = (lane & 1) != 0;
        double acc = (odd ? v1 : v0) + __shfl_xor(odd ? v0 : v1, 1, 64);
#pragma unroll
        for (int off = 2; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
        if (lane < 2) atomicAdd(my + (size_t)t * 2 + lane, acc);
    }
}

// Every other shape up to n = 16, p = 8, m = 2 (chisquare.go:16-95 is shape-generic): the same computation with run-time dimensions on
// lane-private arrays of leading dimension LD (scratch memory: 10-30 KB per lane at LD = 16, as the statement kernels) -- a
// functional path, two orders of magnitude below the fused register kernels; the consistency test of a 9- or 12-state filter design
// is a few thousand runs, not the benchmark.  Same draws, same sums, same order of operations as chisq_kernel.
template <typename T, int LD>
__global__ void __launch_bounds__(64) chisq_gen_kernel(const ChiArgs a, int n, int p, int nc) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    const T *ts = (const T *)a.t_state + tile * ((int64_t)KB_TILE * a.tL.st_elems) + lane;
    const T *tm = (const T *)a.t_model + tile * ((int64_t)KB_TILE * a.tL.mo_elems) + lane;
    const T *ks = (const T *)a.k_state + tile * ((int64_t)KB_TILE * a.kL.st_elems) + lane;
    const T *km = (const T *)a.k_model + tile * ((int64_t)KB_TILE * a.kL.mo_elems) + lane;
    T xt[LD], x[LD], P[LD * LD];
    for (int i = 0; i < n; i++) { xt[i] = ldt(ts, a.tL.st_vec + i); x[i] = ldt(ks, a.kL.st_vec + i); }
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) { const T v = ldt(ks, a.kL.st_mat + symi(i, j)); P[i * LD + j] = v; P[j * LD + i] = v; }
    const uint64_t gfi = (uint64_t)(a.first_run + fi);
    double *my = a.sums + (size_t)(tile % CHI_REPL) * a.nsteps * 2;
    for (int t = 0; t < a.nsteps; t++) {
        T u[2] = {T(0), T(0)};
        if (nc > 0 && a.ncontrols != 1) {
            const T *up = (const T *)a.controls + (int64_t)t * nc;
            for (int c = 0; c < nc; c++) u[c] = up[c];
        }
        // ---- truth: yhat_k = H x_{k-1} + v_k ; x_k = F x_{k-1} [+ G u_k] + w_k  (vanilla.go:138-157, predictionOnly)
        T y[LD], xtn[LD];
        for (int r2 = 0; r2 < p; r2++) {
            T sacc = T(0), v = T(0);
            for (int l = 0; l < n; l++) sacc += ldt(tm, a.tL.mo_H + r2 * n + l) * xt[l];
            for (int k2 = 0; k2 <= r2; k2++) v += ldt(tm, a.tL.mo_LR + symi(k2, r2)) * (T)normal_at(a.seed, gfi, (uint32_t)t, (uint32_t)(a.epoch * 4 + 1), k2);
            y[r2] = sacc + v;
        }
        for (int i = 0; i < n; i++) {
            T sacc = T(0), w = T(0);
            for (int l = 0; l < n; l++) sacc += ldt(tm, a.tL.mo_F + i * n + l) * xt[l];
            if (nc > 0) {
                T g = T(0);
                for (int c = 0; c < nc; c++) g += ldt(tm, a.tL.mo_G + i * nc + c) * u[c];
                sacc = sacc + g;
            }
            for (int k2 = 0; k2 <= i; k2++) w += ldt(tm, a.tL.mo_LQ + symi(k2, i)) * (T)normal_at(a.seed, gfi, (uint32_t)t, (uint32_t)(a.epoch * 4 + 0), k2);
            xtn[i] = sacc + w;
        }
        for (int i = 0; i < n; i++) xt[i] = xtn[i];
        // ---- filter: Vanilla.Update(y, u), Noiseless (vanilla.go:128-220)
        T xm[LD], FP[LD * LD], Pm[LD * LD];
        for (int i = 0; i < n; i++) {
            T sacc = T(0);
            for (int l = 0; l < n; l++) sacc += ldt(km, a.kL.mo_F + i * n + l) * x[l];
            if (nc > 0) {
                T g = T(0);
                for (int c = 0; c < nc; c++) g += ldt(km, a.kL.mo_G + i * nc + c) * u[c];
                sacc = sacc + g;
            }
            xm[i] = sacc;
        }
        for (int i = 0; i < n; i++)
            for (int k2 = 0; k2 < n; k2++) {
                T sacc = T(0);
                for (int l = 0; l < n; l++) sacc += ldt(km, a.kL.mo_F + i * n + l) * P[l * LD + k2];
                FP[i * LD + k2] = sacc;
            }
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) {
                T sacc = T(0);
                for (int k2 = 0; k2 < n; k2++) sacc += FP[i * LD + k2] * ldt(km, a.kL.mo_F + j * n + k2);
                const T v = sacc + ldt(km, a.kL.mo_Q + symi(i, j));
                Pm[i * LD + j] = v; Pm[j * LD + i] = v;   // (the upper triangle, mirrored: what chisq_kernel's packed Pm is)
            }
        T PHt[LD * 8], S[8 * 8], Si[8 * 8], K[LD * 8];
        for (int i = 0; i < n; i++)
            for (int c = 0; c < p; c++) {
                T sacc = T(0);
                for (int l = 0; l < n; l++) sacc += Pm[i * LD + l] * ldt(km, a.kL.mo_H + c * n + l);
                PHt[i * 8 + c] = sacc;
            }
        for (int r2 = 0; r2 < p; r2++)
            for (int c = 0; c < p; c++) {
                T sacc = T(0);
                for (int i = 0; i < n; i++) sacc += ldt(km, a.kL.mo_H + r2 * n + i) * PHt[i * 8 + c];
                S[r2 * 8 + c] = sacc + ldt(km, a.kL.mo_R + symi(r2 < c ? r2 : c, r2 < c ? c : r2));
            }
        {
            T Sw[8 * 8];
            for (int e = 0; e < 64; e++) Sw[e] = S[e];
            (void)inverse_lu_rt<T, 8>(p, Sw, Si);
        }
        for (int i = 0; i < n; i++)
            for (int c = 0; c < p; c++) {
                T sacc = T(0);
                for (int k2 = 0; k2 < p; k2++) sacc += PHt[i * 8 + k2] * Si[k2 * 8 + c];
                K[i * 8 + c] = sacc;
            }
        T innov[8];
        for (int r2 = 0; r2 < p; r2++) {
            T sacc = T(0);
            for (int l = 0; l < n; l++) sacc += ldt(km, a.kL.mo_H + r2 * n + l) * xm[l];
            innov[r2] = y[r2] - sacc;
        }
        for (int i = 0; i < n; i++) {
            T sacc = T(0);
            for (int c = 0; c < p; c++) sacc += K[i * 8 + c] * innov[c];
            x[i] = xm[i] + sacc;
        }
        // Joseph form: P+ = A P- A^T + K R K^T, A = I - K H (upper triangle, mirrored)
        T A[LD * LD], AP[LD * LD];
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                T sacc = T(0);
                for (int c = 0; c < p; c++) sacc += K[i * 8 + c] * ldt(km, a.kL.mo_H + c * n + j);
                A[i * LD + j] = (i == j ? T(1) : T(0)) - sacc;
            }
        for (int i = 0; i < n; i++)
            for (int k2 = 0; k2 < n; k2++) {
                T sacc = T(0);
                for (int l = 0; l < n; l++) sacc += A[i * LD + l] * Pm[l * LD + k2];
                AP[i * LD + k2] = sacc;
            }
        for (int i = 0; i < n; i++) {
            T kr[8];
            for (int c = 0; c < p; c++) {
                T sacc = T(0);
                for (int k2 = 0; k2 < p; k2++) sacc += K[i * 8 + k2] * ldt(km, a.kL.mo_R + symi(k2 < c ? k2 : c, k2 < c ? c : k2));
                kr[c] = sacc;
            }
            for (int j = i; j < n; j++) {
                T sacc = T(0), s2 = T(0);
                for (int k2 = 0; k2 < n; k2++) sacc += AP[i * LD + k2] * A[j * LD + k2];
                for (int c = 0; c < p; c++) s2 += kr[c] * K[j * 8 + c];
                const T v = sacc + s2;
                P[i * LD + j] = v; P[j * LD + i] = v;
            }
        }
        // ---- statistics (chisquare.go:46-77)
        double nis = 0.0, nees = 0.0;
        if (a.with_nis) {
            T sacc = T(0);
            for (int r2 = 0; r2 < p; r2++) {
                T v = T(0);
                for (int c = 0; c < p; c++) v += Si[r2 * 8 + c] * innov[c];
                sacc += innov[r2] * v;
            }
            nis = (double)sacc;
        }
        if (a.with_nees) {
            T Pw[LD * LD], Pi[LD * LD], dlt[LD];
            for (int i = 0; i < n; i++) {
                dlt[i] = xt[i] - x[i];
                for (int j = 0; j < n; j++) Pw[i * LD + j] = P[i * LD + j];
            }
            (void)inverse_lu_rt<T, LD>(n, Pw, Pi);
            T sacc = T(0);
            for (int i = 0; i < n; i++) {
                T v = T(0);
                for (int j = 0; j < n; j++) v += Pi[i * LD + j] * dlt[j];
                sacc += dlt[i] * v;
            }
            nees = (double)sacc;
        }
        const double v0 = active ? nis : 0.0, v1 = active ? nees : 0.0;
        const bool odd = (lane & 1) != 0;
        double acc = (odd ? v1 : v0) + __shfl_xor(odd ? v0 : v1, 1, 64);
#pragma unroll
        for (int off = 2; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
        if (lane < 2) atomicAdd(my + (size_t)t * 2 + lane, acc);
    }
}

// =====================================================================================
// Every other shape with ONE filter fanned out (round 5; chisquare.go:16-95 is shape-generic and NewChiSquare takes ONE kf and the runs
// of ONE Monte-Carlo filter): the covariance recursion of the tested filter does not see the measurements -- P-, S, K, P+ and the two
// inverses the statistics need are the same for every run, bit for bit, because every run performs the same operations on the same
// numbers.  chisq_cov_kernel runs that recursion ONCE (one workgroup, matrices in LDS, chisq_gen_kernel's sums in its order) and writes
// per step K | inverse(P+) | inverse(H P- H^T + R); chisq_shared_kernel then advances one run per lane with only the truth state and the
// filter state in registers (2 n doubles), the models in LDS as broadcast operands (mc_gen_kernel's way) and the step's table entries
// read at wave-uniform addresses: ~3.5 n^2 FMAs and the draws per run-step where the whole Update takes ~10 n^3.
// =====================================================================================
// inverse(M) by Gauss-Jordan elimination with partial pivoting (first largest entry of the column, as dgetf2 picks), the whole workgroup on
// matrices in LDS: M (leading dimension ld) is destroyed, X receives the inverse.  (One thread running the statement kernels' LU routine
// on private arrays took 0.35 ms per step at 12 states -- a dependent chain through scratch memory; this takes microseconds.)
template <typename T>
__device__ void lds_inverse(int n, int ld, T *M, T *X, T *col, int *pivot) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < n * n; e += nt) X[(e / n) * ld + e % n] = (e / n == e % n) ? T(1) : T(0);
    for (int k = 0; k < n; k++) {
        __syncthreads();
        if (tid == 0) {
            int piv = k;
            T best = fabs(M[k * ld + k]);
            for (int i = k + 1; i < n; i++) {
                const T v = fabs(M[i * ld + k]);
                if (v > best) { best = v; piv = i; }
            }
            *pivot = piv;
        }
        __syncthreads();
        const int piv = *pivot;
        if (piv != k) {
            for (int c = tid; c < 2 * n; c += nt) {
                T *A = c < n ? M : X;
                const int cc = c < n ? c : c - n;
                const T t0 = A[k * ld + cc], t1 = A[piv * ld + cc];
                A[k * ld + cc] = t1; A[piv * ld + cc] = t0;
            }
        }
        __syncthreads();
        const T rinv = T(1) / M[k * ld + k];
        for (int i = tid; i < n; i += nt) col[i] = M[i * ld + k];   // the column being eliminated, before anybody rewrites it
        __syncthreads();
        for (int c = tid; c < 2 * n; c += nt) {   // the pivot row, normalised
            T *A = c < n ? M : X;
            const int cc = c < n ? c : c - n;
            A[k * ld + cc] = A[k * ld + cc] * rinv;
        }
        __syncthreads();
        for (int e = tid; e < n * 2 * n; e += nt) {
            const int i = e / (2 * n), c = e % (2 * n);
            if (i == k) continue;
            T *A = c < n ? M : X;
            const int cc = c < n ? c : c - n;
            A[i * ld + cc] -= col[i] * A[k * ld + cc];
        }
    }
    __syncthreads();
}

template <typename T, int LD>
__global__ void __launch_bounds__(256) chisq_cov_kernel(const ChiArgs a, int n, int p, T *__restrict__ table) {
    constexpr int PM = 8;
    __shared__ T sP[LD * LD], sPm[LD * LD], sFP[LD * LD], sF[LD * LD], sQ[LD * LD], sA[LD * LD], sAP[LD * LD], sH[PM * LD], sR[PM * PM], sPHt[LD * PM], sS[PM * PM], sSi[PM * PM],
        sK[LD * PM], sKR[LD * PM], sPi[LD * LD], sCol[LD];
    __shared__ int sPiv;
    const int tid = threadIdx.x, nt = blockDim.x;
    const T *ks = (const T *)a.k_state, *km = (const T *)a.k_model;   // run 0 of tile 0: element e at [e * KB_TILE]
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e % n;
        sP[i * LD + j] = ldt(ks, a.kL.st_mat + symi(i < j ? i : j, i < j ? j : i));
        sF[i * LD + j] = ldt(km, a.kL.mo_F + i * n + j);
        sQ[i * LD + j] = ldt(km, a.kL.mo_Q + symi(i < j ? i : j, i < j ? j : i));
    }
    for (int e = tid; e < p * n; e += nt) sH[(e / n) * LD + e % n] = ldt(km, a.kL.mo_H + e);
    for (int e = tid; e < p * p; e += nt) { const int r = e / p, c = e % p; sR[r * PM + c] = ldt(km, a.kL.mo_R + symi(r < c ? r : c, r < c ? c : r)); }
    __syncthreads();
    const int ts = n * p + n * n + p * p;
    for (int t = 0; t < a.nsteps; t++) {
        for (int e = tid; e < n * n; e += nt) {   // FP = F P
            const int i = e / n, k2 = e % n;
            T sacc = T(0);
            for (int l = 0; l < n; l++) sacc += sF[i * LD + l] * sP[l * LD + k2];
            sFP[i * LD + k2] = sacc;
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) {   // P- = FP F^T + Q: the upper triangle, mirrored
            const int i = e / n, j = e % n;
            if (j >= i) {
                T sacc = T(0);
                for (int k2 = 0; k2 < n; k2++) sacc += sFP[i * LD + k2] * sF[j * LD + k2];
                const T v = sacc + sQ[i * LD + j];
                sPm[i * LD + j] = v; sPm[j * LD + i] = v;
            }
        }
        __syncthreads();
        for (int e = tid; e < n * p; e += nt) {   // P- H^T
            const int i = e / p, c = e % p;
            T sacc = T(0);
            for (int l = 0; l < n; l++) sacc += sPm[i * LD + l] * sH[c * LD + l];
            sPHt[i * PM + c] = sacc;
        }
        __syncthreads();
        for (int e = tid; e < p * p; e += nt) {   // S = H P- H^T + R
            const int r2 = e / p, c = e % p;
            T sacc = T(0);
            for (int i = 0; i < n; i++) sacc += sH[r2 * LD + i] * sPHt[i * PM + c];
            sS[r2 * PM + c] = sacc + sR[r2 * PM + c];
        }
        __syncthreads();
        lds_inverse<T>(p, PM, sS, sSi, sCol, &sPiv);   // (S is not needed again: P+ uses K R K^T)
        for (int e = tid; e < n * p; e += nt) {   // K = P- H^T S^-1
            const int i = e / p, c = e % p;
            T sacc = T(0);
            for (int k2 = 0; k2 < p; k2++) sacc += sPHt[i * PM + k2] * sSi[k2 * PM + c];
            sK[i * PM + c] = sacc;
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) {   // A = I - K H
            const int i = e / n, j = e % n;
            T sacc = T(0);
            for (int c = 0; c < p; c++) sacc += sK[i * PM + c] * sH[c * LD + j];
            sA[i * LD + j] = (i == j ? T(1) : T(0)) - sacc;
        }
        for (int e = tid; e < n * p; e += nt) {   // K R
            const int i = e / p, c = e % p;
            T sacc = T(0);
            for (int k2 = 0; k2 < p; k2++) sacc += sK[i * PM + k2] * sR[k2 * PM + c];
            sKR[i * PM + c] = sacc;
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) {   // A P-
            const int i = e / n, k2 = e % n;
            T sacc = T(0);
            for (int l = 0; l < n; l++) sacc += sA[i * LD + l] * sPm[l * LD + k2];
            sAP[i * LD + k2] = sacc;
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) {   // P+ = A P- A^T + K R K^T: the upper triangle, mirrored
            const int i = e / n, j = e % n;
            if (j >= i) {
                T sacc = T(0), s2 = T(0);
                for (int k2 = 0; k2 < n; k2++) sacc += sAP[i * LD + k2] * sA[j * LD + k2];
                for (int c = 0; c < p; c++) s2 += sKR[i * PM + c] * sK[j * PM + c];
                const T v = sacc + s2;
                sP[i * LD + j] = v; sP[j * LD + i] = v;
            }
        }
        __syncthreads();
        if (a.with_nees) {   // inverse(P+) (chisquare.go:50-51), on a copy: P+ carries on
            for (int e = tid; e < n * n; e += nt) sFP[(e / n) * LD + e % n] = sP[(e / n) * LD + e % n];
            __syncthreads();
            lds_inverse<T>(n, LD, sFP, sPi, sCol, &sPiv);
        }
        T *row = table + (size_t)t * ts;
        for (int e = tid; e < n * p; e += nt) row[e] = sK[(e / p) * PM + e % p];
        for (int e = tid; e < n * n; e += nt) row[n * p + e] = a.with_nees ? sPi[(e / n) * LD + e % n] : T(0);
        for (int e = tid; e < p * p; e += nt) row[n * p + n * n + e] = sSi[(e / p) * PM + e % p];
        __syncthreads();
    }
}

template <typename T, int NS>
__global__ void __launch_bounds__(256, 2) chisq_shared_kernel(const ChiArgs a, int n, int p, int nc, const T *__restrict__ table) {
    constexpr int TQ = tri(NS), PM = 8, TP = tri(PM);
    __shared__ T sF[NS * NS], sLQ[TQ], sG[NS * 2], sH[PM * NS], sLR[TP], kF[NS * NS], kG[NS * 2], kH[PM * NS];
    {
        const T *tm = (const T *)a.t_model, *km = (const T *)a.k_model;   // run 0's model blocks
        for (int e = threadIdx.x; e < NS * NS; e += blockDim.x) {
            const int i = e / NS, l = e % NS;
            sF[e] = (i < n && l < n) ? ldt(tm, a.tL.mo_F + i * n + l) : T(0);
            kF[e] = (i < n && l < n) ? ldt(km, a.kL.mo_F + i * n + l) : T(0);
        }
        for (int e = threadIdx.x; e < TQ; e += blockDim.x) sLQ[e] = e < tri(n) ? ldt(tm, a.tL.mo_LQ + e) : T(0);
        for (int e = threadIdx.x; e < NS * 2; e += blockDim.x) {
            const int i = e / 2, c = e % 2;
            sG[e] = (i < n && c < nc) ? ldt(tm, a.tL.mo_G + i * nc + c) : T(0);
            kG[e] = (i < n && c < nc) ? ldt(km, a.kL.mo_G + i * nc + c) : T(0);
        }
        for (int e = threadIdx.x; e < PM * NS; e += blockDim.x) {
            const int r = e / NS, l = e % NS;
            sH[e] = (r < p && l < n) ? ldt(tm, a.tL.mo_H + r * n + l) : T(0);
            kH[e] = (r < p && l < n) ? ldt(km, a.kL.mo_H + r * n + l) : T(0);
        }
        for (int e = threadIdx.x; e < TP; e += blockDim.x) sLR[e] = e < tri(p) ? ldt(tm, a.tL.mo_LR + e) : T(0);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    const T *ts0 = (const T *)a.t_state, *ks0 = (const T *)a.k_state;   // every run starts from run 0's initial estimate (one filter fanned out)
    T xt[NS], x[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) { xt[i] = i < n ? ldt(ts0, a.tL.st_vec + i) : T(0); x[i] = i < n ? ldt(ks0, a.kL.st_vec + i) : T(0); }
    const uint64_t gfi = (uint64_t)(a.first_run + fi);
    double *my = a.sums + (size_t)(tile % CHI_REPL) * a.nsteps * 2;
    const int tstride = n * p + n * n + p * p;
    for (int t = 0; t < a.nsteps; t++) {
        asm volatile("" ::: "memory");   // (the models are re-read from LDS every step, not hoisted into hundreds of registers)
        const T *row = table + (size_t)t * tstride;
        T u0 = T(0), u1 = T(0);
        if (nc > 0 && a.ncontrols != 1) {
            const T *up = (const T *)a.controls + (int64_t)t * nc;
            u0 = up[0];
            u1 = nc > 1 ? up[1] : T(0);
        }
        // ---- truth: yhat_k = H x_{k-1} + v_k ; x_k = F x_{k-1} [+ G u_k] + w_k  (vanilla.go:138-157, predictionOnly)
        T zv[PM], y[PM];
#pragma unroll
        for (int k2 = 0; k2 < PM; k2 += 2) {
            zv[k2] = T(0); zv[k2 + 1] = T(0);
            if (k2 < p) {   // (wave-uniform)
                uint32_t r[4];
                Philox::gen(a.seed, gfi, (uint32_t)t, ((uint32_t)(a.epoch * 4 + 1) << 8) | (uint32_t)(k2 >> 1), r);
                double z0, z1;
                box_muller(r, z0, z1);
                zv[k2] = (T)z0; zv[k2 + 1] = (T)z1;
            }
        }
#pragma unroll
        for (int r2 = 0; r2 < PM; r2++) {
            T sacc = T(0), v = T(0);
            if (r2 < p) {
#pragma unroll
                for (int l = 0; l < NS; l++) sacc += sH[r2 * NS + l] * xt[l];
#pragma unroll
                for (int k2 = 0; k2 <= r2; k2++) v += sLR[symi(k2, r2)] * zv[k2];
            }
            y[r2] = sacc + v;
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);   // (phase by phase: interleaved, the broadcast operands of all phases are alive together)
        T xtn[NS], z[NS];
#pragma unroll
        for (int k2 = 0; k2 < NS; k2 += 2) {
            z[k2] = T(0);
            if (k2 + 1 < NS) z[k2 + 1] = T(0);
            if (k2 < n) {
                uint32_t r[4];
                Philox::gen(a.seed, gfi, (uint32_t)t, ((uint32_t)(a.epoch * 4 + 0) << 8) | (uint32_t)(k2 >> 1), r);
                double z0, z1;
                box_muller(r, z0, z1);
                z[k2] = (T)z0;
                if (k2 + 1 < NS) z[k2 + 1] = (T)z1;
            }
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T sacc = T(0), w = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) sacc += sF[i * NS + l] * xt[l];
            if (nc > 0) {
                T g = T(0);
                g += sG[i * 2 + 0] * u0;
                if (nc > 1) g += sG[i * 2 + 1] * u1;
                sacc = sacc + g;
            }
#pragma unroll
            for (int k2 = 0; k2 <= i; k2++) w += sLQ[symi(k2, i)] * z[k2];
            xtn[i] = sacc + w;
            pin(xtn[i]);
            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two rows of broadcast operands in flight at a time
        }
#pragma unroll
        for (int i = 0; i < NS; i++) xt[i] = xtn[i];
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);   // (phase by phase: interleaved, the broadcast operands of all phases are alive together)
        // ---- filter: the state half of Vanilla.Update(y, u), Noiseless (vanilla.go:138-146, :183-195); K from the table
        T xm[NS], innov[PM];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T sacc = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) sacc += kF[i * NS + l] * x[l];
            if (nc > 0) {
                T g = T(0);
                g += kG[i * 2 + 0] * u0;
                if (nc > 1) g += kG[i * 2 + 1] * u1;
                sacc = sacc + g;
            }
            xm[i] = sacc;
            pin(xm[i]);
            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two rows of broadcast operands in flight at a time
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);   // (phase by phase: interleaved, the broadcast operands of all phases are alive together)
#pragma unroll
        for (int r2 = 0; r2 < PM; r2++) {
            T sacc = T(0);
            if (r2 < p) {
#pragma unroll
                for (int l = 0; l < NS; l++) sacc += kH[r2 * NS + l] * xm[l];
            }
            innov[r2] = r2 < p ? y[r2] - sacc : T(0);
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T sacc = T(0);
            if (i < n)
                for (int c = 0; c < p; c++) sacc += row[i * p + c] * innov[c];
            x[i] = xm[i] + sacc;
            pin(x[i]);
            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two rows of broadcast operands in flight at a time
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);   // (phase by phase: interleaved, the broadcast operands of all phases are alive together)
        // ---- statistics (chisquare.go:46-77)
        double nis = 0.0, nees = 0.0;
        if (a.with_nis) {
            const T *si = row + n * p + n * n;
            T sacc = T(0);
            for (int r2 = 0; r2 < p; r2++) {
                T v = T(0);
                for (int c = 0; c < p; c++) v += si[r2 * p + c] * innov[c];
                sacc += innov[r2] * v;
            }
            nis = (double)sacc;
        }
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);   // (phase by phase: interleaved, the broadcast operands of all phases are alive together)
        if (a.with_nees) {
            const T *pi = row + n * p;
            T dlt[NS];
#pragma unroll
            for (int i = 0; i < NS; i++) dlt[i] = xt[i] - x[i];
            T sacc = T(0);
#pragma unroll
            for (int i = 0; i < NS; i++) {
                if (i < n) {
                    T v = T(0);
#pragma unroll
                    for (int j = 0; j < NS; j++)
                        if (j < n) v += pi[i * n + j] * dlt[j];
                    sacc += dlt[i] * v;
                    pin(sacc);
                }
                if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two rows of broadcast operands in flight at a time
            }
            nees = (double)sacc;
        }
        const double v0 = active ? nis : 0.0, v1 = active ? nees : 0.0;
        const bool odd = (lane & 1) != 0;
        double acc = (odd ? v1 : v0) + __shfl_xor(odd ? v0 : v1, 1, 64);
#pragma unroll
        for (int off = 2; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
        if (lane < 2) atomicAdd(my + (size_t)t * 2 + lane, acc);
    }
}

template <typename T, int NS, int NM>
static bool chi_try(const Batch &tb, const ChiArgs &a, int n, int p, int nc) {
    if (n != NS || p != NM) return false;
    const dim3 grid = tile_grid(a.ntiles), block(256);
    switch (nc) {
    case 0: hipLaunchKernelGGL((chisq_kernel<T, NS, NM, 0>), grid, block, 0, tb.stream, a); return true;
    case 1: hipLaunchKernelGGL((chisq_kernel<T, NS, NM, 1>), grid, block, 0, tb.stream, a); return true;
    case 2: hipLaunchKernelGGL((chisq_kernel<T, NS, NM, 2>), grid, block, 0, tb.stream, a); return true;
    }
    return false;
}

int chi_repl() { return CHI_REPL; }

int launch_chisq(const Batch &tb, const ChiArgs &a, int n, int p, int nc, void *shared_table) {
    bool ok = false;
    static const bool shared_all = [] { const char *e = getenv("KB_CHISQ_SHARED_ALL"); return e && *e && *e != '0'; }();   // (diagnostic: the fused shapes on the shared-covariance path)
    // (6,3) with ONE filter fanned out runs the shared-covariance path as well: 10.6 G run-steps/s against 7.5 G on the fused kernel, which
    // holds a whole Vanilla update per lane in 512 registers at one wave per SIMD (256k runs x 200 steps); the smaller fused shapes stay
    // ((4,2): 22.3 G fused against 24.0 G shared at 1M runs x 1086 steps, 21.2 against 16.5 at 256k x 200 -- the serial recursion's share)
    // ... from ~128k runs on: the recursion is SERIAL in one workgroup (~10 us per step whatever the ensemble), the fused kernel's step costs
    // N / 7.5 G -- below ~160k runs the fused kernel is ahead ((4,2): 21.2 G fused against 16.5 G shared at 256k x 200)
    const bool shared_63 = shared_table && n == 6 && p == 3 && a.N >= (int64_t(1) << 17);
    if (tb.dtype == KB_F64 && !(shared_all && shared_table) && !shared_63)
        ok = chi_try<double, 2, 1>(tb, a, n, p, nc) || chi_try<double, 3, 1>(tb, a, n, p, nc) || chi_try<double, 4, 2>(tb, a, n, p, nc) ||
             chi_try<double, 6, 3>(tb, a, n, p, nc);
    if (!ok && tb.dtype == KB_F64 && n <= 16 && p <= 8 && nc <= 2 && shared_table) {
        // ONE filter fanned out (no per-run model, no per-run initial estimate): the covariance recursion once, then one run per lane
        if (n <= 8) hipLaunchKernelGGL((chisq_cov_kernel<double, 8>), dim3(1), dim3(256), 0, tb.stream, a, n, p, (double *)shared_table);
        else hipLaunchKernelGGL((chisq_cov_kernel<double, 16>), dim3(1), dim3(256), 0, tb.stream, a, n, p, (double *)shared_table);
        const dim3 grid = tile_grid(a.ntiles), block(256);
        if (n <= 4) hipLaunchKernelGGL((chisq_shared_kernel<double, 4>), grid, block, 0, tb.stream, a, n, p, nc, (const double *)shared_table);
        else if (n <= 6) hipLaunchKernelGGL((chisq_shared_kernel<double, 6>), grid, block, 0, tb.stream, a, n, p, nc, (const double *)shared_table);
        else if (n <= 8) hipLaunchKernelGGL((chisq_shared_kernel<double, 8>), grid, block, 0, tb.stream, a, n, p, nc, (const double *)shared_table);
        else if (n <= 12) hipLaunchKernelGGL((chisq_shared_kernel<double, 12>), grid, block, 0, tb.stream, a, n, p, nc, (const double *)shared_table);
        else hipLaunchKernelGGL((chisq_shared_kernel<double, 16>), grid, block, 0, tb.stream, a, n, p, nc, (const double *)shared_table);
        ok = true;
    }
    if (!ok && tb.dtype == KB_F64 && n <= 16 && p <= 8 && nc <= 2) {   // per-run models or initial estimates: run-time dimensions on lane-private arrays
        const HeavyScope hs(tb, n > 8);   // LD = 16: scratch-heavy, see kb_internal.h
        if (n <= 8) hipLaunchKernelGGL((chisq_gen_kernel<double, 8>), dim3((unsigned)a.ntiles), dim3(64), 0, hs.stream, a, n, p, nc);
        else hipLaunchKernelGGL((chisq_gen_kernel<double, 16>), dim3((unsigned)a.ntiles), dim3(64), 0, hs.stream, a, n, p, nc);
        ok = true;
    }
    if (!ok) {
        set_error("kb_chisquare: no kernel for n=%d p=%d m=%d (fp64, n <= 16, p <= 8, m <= 2)", n, p, nc);
        return KB_ERR_UNSUPPORTED;
    }
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb

using namespace kb;

extern "C" int kb_chisquare(kb_batch *truth, kb_batch *kf, int steps, const double *controls, int ncontrols, int64_t first_run,
                            int replay_last_mc, int with_nees, int with_nis, double *sums) {
    if (!truth || !kf || !sums) { set_error("null argument"); return KB_ERR_INVALID; }
    double *d_folded = nullptr;
    int rc = chisq_run_device(*truth, *kf, steps, controls, ncontrols, first_run, replay_last_mc, with_nees, with_nis, &d_folded);
    if (rc) return rc;
    KB_HIP(hipMemcpyAsync(sums, d_folded, (size_t)steps * 2 * sizeof(double), hipMemcpyDeviceToHost, truth->stream));   // [steps][2]: sum NIS, sum NEES
    KB_HIP(hipStreamSynchronize(truth->stream));
    return KB_OK;
}

int kb::chisq_run_device(Batch &tb, Batch &kb_, int steps, const double *controls, int ncontrols, int64_t first_run, int replay_last_mc,
                         int with_nees, int with_nis, double **folded) {
    Batch *truth = &tb, *kf = &kb_;
    if (!with_nees && !with_nis) { set_error("Chi Square requires either NEES or NIS or both"); return KB_ERR_INVALID; }  // chisquare.go:17-19
    if (!truth->initialized || !kf->initialized) { set_error("kb_init has not been called"); return KB_ERR_INVALID; }
    if (truth->kind != KB_VANILLA_PREDICT || truth->noise_kind != KB_NOISE_AWGN) {
        set_error("the Monte-Carlo truth must be a pure-predictor Vanilla batch with AWGN noise");
        return KB_ERR_INVALID;
    }
    if (kf->kind != KB_VANILLA) { set_error("the tested filter must be a Vanilla batch"); return KB_ERR_UNSUPPORTED; }
    if (truth->N != kf->N || truth->n != kf->n || truth->p != kf->p || truth->m != kf->m || truth->dtype != kf->dtype || truth->device != kf->device) {
        set_error("truth and filter batches must agree in size, shape, dtype and device");
        return KB_ERR_DIMS;
    }
    if (steps < 1) { set_error("steps must be >= 1"); return KB_ERR_INVALID; }
    if (ncontrols != 1 && ncontrols != steps) {  // chisquare.go:27-36
        set_error("must provide as much control vectors as steps, or just one control vector");
        return KB_ERR_INVALID;
    }
    if (truth->need_ctrl != kf->need_ctrl) { set_error("truth and filter disagree on needCtrl"); return KB_ERR_DIMS; }
    int rc = use_device(*truth);
    if (rc) return rc;
    const int m = truth->m;
    if (truth->need_ctrl) {
        if (!controls) { set_error("controls required (needCtrl)"); return KB_ERR_INVALID; }
        const size_t cnt = (size_t)ncontrols * m, bytes = cnt * truth->esize();
        if (truth->ctrl_bytes < bytes) {
            if (truth->d_ctrl) KB_HIP(dev_free(truth->d_ctrl));
            truth->d_ctrl = nullptr; truth->ctrl_bytes = 0;
            KB_HIP(dev_alloc(&truth->d_ctrl, bytes));
            truth->ctrl_bytes = bytes;
        }
        KB_HIP(hipMemcpy(truth->d_ctrl, controls, bytes, hipMemcpyHostToDevice));  // fp64 only (see launch_chisq)
    }
    const int repl = chi_repl();
    const size_t nrep = (size_t)repl * steps * 2, ndbl = nrep + (size_t)steps * 2;   // [repl][steps][2] | folded [steps][2]
    if (truth->mc_bytes < ndbl * sizeof(double)) {
        if (truth->d_mc) KB_HIP(dev_free(truth->d_mc));
        truth->d_mc = nullptr; truth->mc_bytes = 0;
        KB_HIP(dev_alloc((void **)&truth->d_mc, ndbl * sizeof(double)));
        truth->mc_bytes = ndbl * sizeof(double);
    }
    KB_HIP(hipStreamSynchronize(kf->stream));
    KB_HIP(hipMemsetAsync(truth->d_mc, 0, ndbl * sizeof(double), truth->stream));
    ChiArgs a;
    memset(&a, 0, sizeof(a));
    a.t_state = truth->d_state0; a.t_model = truth->d_model; a.k_state = kf->d_state0; a.k_model = kf->d_model;  // kf.Reset() per run (chisquare.go:39)
    a.tL = truth->L; a.kL = kf->L;
    a.N = truth->N; a.ntiles = truth->ntiles; a.first_run = first_run;
    a.epoch = (replay_last_mc && truth->epoch > 0) ? truth->epoch - 1 : truth->epoch;
    a.seed = truth->seed;
    a.nsteps = steps; a.ncontrols = ncontrols; a.need_ctrl = truth->need_ctrl; a.with_nees = with_nees; a.with_nis = with_nis;
    a.controls = truth->d_ctrl; a.sums = truth->d_mc;
    // the table of the shared-covariance path (chisq_cov_kernel): [steps][n p + n n + p p] doubles behind the sums; only when both
    // batches are ONE filter fanned out (kb_replicate / broadcast uploads: no per-run model field, no per-run x0 / P0)
    void *table = nullptr;
    if (!truth->per_filter_model && !truth->per_filter_init && !kf->per_filter_model && !kf->per_filter_init && truth->dtype == KB_F64) {
        const size_t tb_bytes = (size_t)steps * (size_t)(truth->n * truth->p + truth->n * truth->n + truth->p * truth->p) * sizeof(double);
        if (truth->chi_table_bytes < tb_bytes) {
            if (truth->d_chi_table) KB_HIP(dev_free(truth->d_chi_table));
            truth->d_chi_table = nullptr; truth->chi_table_bytes = 0;
            KB_HIP(dev_alloc(&truth->d_chi_table, tb_bytes));
            truth->chi_table_bytes = tb_bytes;
        }
        table = truth->d_chi_table;
    }
    if ((rc = launch_chisq(*truth, a, truth->n, truth->p, truth->need_ctrl ? m : 0, table))) return rc;
    *folded = truth->d_mc + nrep;
    if ((rc = launch_fold(truth->stream, truth->d_mc, repl, (int64_t)steps * 2, *folded))) return rc;
    if (!replay_last_mc) truth->epoch++;
    return KB_OK;
}
