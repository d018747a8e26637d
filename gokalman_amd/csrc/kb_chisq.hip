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

int launch_chisq(const Batch &tb, const ChiArgs &a, int n, int p, int nc) {
    bool ok = false;
    if (tb.dtype == KB_F64)
        ok = chi_try<double, 2, 1>(tb, a, n, p, nc) || chi_try<double, 3, 1>(tb, a, n, p, nc) || chi_try<double, 4, 2>(tb, a, n, p, nc) ||
             chi_try<double, 6, 3>(tb, a, n, p, nc);
    if (!ok && tb.dtype == KB_F64 && n <= 16 && p <= 8 && nc <= 2) {   // every other shape: run-time dimensions on lane-private arrays
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
    if ((rc = launch_chisq(*truth, a, truth->n, truth->p, truth->need_ctrl ? m : 0))) return rc;
    *folded = truth->d_mc + nrep;
    if ((rc = launch_fold(truth->stream, truth->d_mc, repl, (int64_t)steps * 2, *folded))) return rc;
    if (!replay_last_mc) truth->epoch++;
    return KB_OK;
}
