// kb_srif_reg.hip -- register-resident SRIF update (srif.go:101-160, :298-340, helper.go:142-172)
// for the benchmark shape (n = 12, p = 6, fp32) and the reference tests' shape (n = 6, p = 2).
// One filter per lane.  Phi / H-tilde are read in place from the caller's planar arrays after
// kb_prepare_dev (zero-copy) or from the model block after kb_prepare.
//
// Two launches per Update, mirroring the reference's own split: the time update
// (srif.go:111-141, also the whole of Predict()) rewrites (b, R) as (bBar, RBar) in place, the
// measurement update (srif.go:143-156, :298-340) runs the 18 x 13 Householder panel on it (1 wave/SIMD,
// 512-register budget).  RBar (164 MB at 256k filters) stays in the Infinity Cache between the two.
// Differences from the statement-by-statement generic kernel (rounding level only):
//   State(prev) = R^-1 b and RBar = R Phi^-1 are obtained by LU solves instead of
//   inverse-then-multiply (srif.go:111-115, :223-234); only exact singularity / non-finite
//   results are flagged (the generic kernel also applies gonum's cond > 1e16 test).
// Failure semantics, the same on every SRIF path (fused, time + meas, generic) and the reference's (srif.go:111-114:
// `return nil, err` before anything is assigned): a filter whose Phi (or R, in State(prev)) is singular at step k gets
// KB_ST_SINGULAR and keeps its estimate for THAT step only; step k+1 runs normally (the status word is a sticky
// report, not a gate).  A non-finite Householder result is stored as it is -- helper.go:142-172 has no guard -- and
// flagged KB_ST_NONFINITE.
// Algorithmic bytes per filter-step: b 12 + R 144 + Phi 144 + Htilde 72 +
// L 36 + real 6 + computed 6 read, b 12 + R 144 written = 576 elements = 2304 B in fp32.
#include <cstdlib>

#include "kb_internal.h"
#include "kb_static.h"

namespace kb {


__device__ __forceinline__ int nib(uint64_t perm, int r) { return (int)((perm >> (4 * r)) & 15u); }

// ---- time update (srif.go:111-141): b <- bBar, R <- RBar = R Phi^-1, in place -------------------
// One wave alone on a SIMD issues a VALU instruction every 4 cycles, two waves every 2 (MI355X_MICROARCH.md,
// "vector-instruction ISSUE cost"), and this kernel is VALU-paced (~4.5k instructions per filter tile), so it is
// written to fit 256 registers and a few KB of LDS per wave: Phi is factorised in VGPRs (row exchanges by
// select, skipped wave-uniformly when no lane needs one), the factors stay there for the 12 row solves, and LDS
// is only the scatter buffer that undoes the row permutation (24 floats per lane).
template <typename T, int NS, bool FULL, bool EXT>
__global__ void __launch_bounds__(256, (sizeof(T) * NS * NS > 600 ? 1 : 2)) srif_time_kernel(const StepArgs a) {
    constexpr int RG = 2;
    __shared__ T lds[4 * RG * NS * KB_TILE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (NS + NS * NS)) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *ephi = EXT ? (const T *)a.ext_phi + (active ? fi : 0) : nullptr;
    T *ltmp = lds + wv * (RG * NS * KB_TILE) + lane;
    unsigned err = 0;
    T xprev[NS];
    {   // State(prev) = R^-1 b (srif.go:223-234)
        T Rw[NS * NS];
#pragma unroll
        for (int i = 0; i < NS; i++) xprev[i] = ldt(st, i);
#pragma unroll
        for (int e = 0; e < NS * NS; e++) Rw[e] = ldt(st, NS + e);
        // R is upper triangular whenever the last writer was a measurement update (srif.go:334-337 zeroes the
        // sub-columns) or the constructor (diagonal R0): pivoted LU then degenerates to L = I, U = R, no exchanges.
        // Wave-uniform test; the general path stays for R after Predict() / NON_TRI_R.
        bool lower = false;
#pragma unroll
        for (int i = 1; i < NS; i++)
#pragma unroll
            for (int j = 0; j < i; j++) lower = lower || (Rw[i * NS + j] != T(0));
        if (__any(lower)) {
            if (lu_solve_inplace<T, NS, 1>(Rw, xprev)) err |= KB_ST_SINGULAR;
        } else {
#pragma unroll
            for (int i = NS - 1; i >= 0; i--) {
                T sum = xprev[i];
#pragma unroll
                for (int k2 = i + 1; k2 < NS; k2++) sum -= Rw[i * NS + k2] * xprev[k2];
                if (Rw[i * NS + i] == T(0)) err |= KB_ST_SINGULAR;
                xprev[i] = sum * (T(1) / Rw[i * NS + i]);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    T lu[NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {   // xBar = Phi State(prev) (srif.go:118) -> LDS slots, read back in pivoted order below
        T s = T(0);
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const T v = EXT ? __builtin_nontemporal_load(ephi + (int64_t)(i * NS + j) * a.ext_ld) : ldnt(mo, a.L.mo_F + i * NS + j);
            lu[i * NS + j] = v;
            s += v * xprev[j];
        }
        ltmp[i * KB_TILE] = s;
    }
    // P Phi = L U (srif.go:111-114's Inverse = Dgetrf + ...): nibble k of perm = original index of the row now in position k
    uint64_t perm = 0xBA9876543210ull;
#pragma unroll
    for (int j = 0; j < NS; j++) {
#pragma unroll
        for (int r = j + 1; r < NS; r++) {
            const bool sw = fabs(lu[r * NS + j]) > fabs(lu[j * NS + j]);
            if (__any(sw)) {
#pragma unroll
                for (int c = 0; c < NS; c++) {  // whole rows: the L part moves with its row (LAPACK dlaswp)
                    const T t0 = lu[j * NS + c], t1 = lu[r * NS + c];
                    lu[j * NS + c] = sw ? t1 : t0;
                    lu[r * NS + c] = sw ? t0 : t1;
                }
                const uint64_t x = sw ? (((perm >> (4 * j)) ^ (perm >> (4 * r))) & 15u) : 0u;
                perm ^= (x << (4 * j)) | (x << (4 * r));
            }
        }
        const T piv = lu[j * NS + j];
        if (piv == T(0)) err |= KB_ST_SINGULAR;
        const T rp = T(1) / piv;
        lu[j * NS + j] = rp;   // the solves multiply by the reciprocal
#pragma unroll
        for (int r = j + 1; r < NS; r++) {
            const T l = lu[r * NS + j] * rp;
            lu[r * NS + j] = l;
#pragma unroll
            for (int c = j + 1; c < NS; c++) lu[r * NS + c] -= l * lu[j * NS + c];
        }
    }
    // failed: (b, R) stay as they are (srif.go:111-114 returns before any assignment); the measurement kernel of the same
    // Update finds KB_ST_SKIP_STEP, leaves the filter alone and clears the bit again
    if (err) { if (active) atomicOr(a.status + fi, err | (a.predict ? 0u : KB_ST_SKIP_STEP)); return; }
    T *es = FULL ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane : nullptr;
    int poff[NS];  // LDS element offset of original row perm_k
    T xBarP[NS];   // xBar in pivoted order
#pragma unroll
    for (int r = 0; r < NS; r++) {
        poff[r] = nib(perm, r) * KB_TILE;
        xBarP[r] = ltmp[poff[r]];
    }
    // RBar = R Phi^-1, RG rows at a time (independent dependency chains); the next rows of R are prefetched
    // (L2 hits: the state tile was read a moment ago) while the current ones are solved.
    T znext[RG][NS];
#pragma unroll
    for (int g = 0; g < RG; g++)
#pragma unroll
        for (int l = 0; l < NS; l++) znext[g][l] = ldt(st, NS + g * NS + l);
#pragma unroll 1
    for (int i0 = 0; i0 < NS; i0 += RG) {
        T z[RG][NS];
#pragma unroll
        for (int g = 0; g < RG; g++)
#pragma unroll
            for (int l = 0; l < NS; l++) z[g][l] = znext[g][l];
        if (i0 + RG < NS) {
#pragma unroll
            for (int g = 0; g < RG; g++)
#pragma unroll
                for (int l = 0; l < NS; l++) znext[g][l] = ldt(st, NS + (i0 + RG + g) * NS + l);
        }
        // z Phi = r  with  P Phi = L U:  w U = r,  v L = w,  z[perm_k] = v_k   (srif.go:115)
#pragma unroll
        for (int j = 0; j < NS; j++) {
#pragma unroll
            for (int g = 0; g < RG; g++) {
                T sum = z[g][j];
#pragma unroll
                for (int k2 = 0; k2 < j; k2++) sum -= z[g][k2] * lu[k2 * NS + j];
                z[g][j] = sum * lu[j * NS + j];
            }
        }
#pragma unroll
        for (int j = NS - 2; j >= 0; j--) {
#pragma unroll
            for (int g = 0; g < RG; g++) {
                T sum = z[g][j];
#pragma unroll
                for (int k2 = j + 1; k2 < NS; k2++) sum -= z[g][k2] * lu[k2 * NS + j];
                z[g][j] = sum;
            }
        }
#pragma unroll
        for (int g = 0; g < RG; g++) {
            T bb = T(0);
#pragma unroll
            for (int r = 0; r < NS; r++) {
                bb += z[g][r] * xBarP[r];                        // :119 bBar = RBar xBar (same products, pivoted order)
                ltmp[poff[r] + g * NS * KB_TILE] = z[g][r];     // undo the row permutation through LDS
            }
            if (active) stt(st, i0 + g, bb);  // b <- bBar: rows i0.. of R and their b entries are not read again
        }
        if (active) {
#pragma unroll
            for (int g = 0; g < RG; g++)
#pragma unroll
                for (int c = 0; c < NS; c++) {
                    const T v = ltmp[(g * NS + c) * KB_TILE];
                    stt(st, NS + (i0 + g) * NS + c, v);
                    if constexpr (FULL) stt(es, a.L.es_ppred + (i0 + g) * NS + c, v);
                }
        }
    }
}

// ---- measurement update (srif.go:143-156, :298-340): Householder on [[RBar bBar],[L Htilde, L y]] --
template <typename T, int NS, int NM, bool FULL, bool EXT>
__global__ void __launch_bounds__(256, 1) srif_meas_kernel(const StepArgs a) {
    constexpr int COLS = NS + 1;
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    if (active && (a.status[fi] & KB_ST_SKIP_STEP) != 0u) {   // the time update of THIS step failed: the estimate stays as it was
        atomicAnd(a.status + fi, ~KB_ST_SKIP_STEP);
        return;
    }
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (NS + NS * NS)) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *eh = EXT ? (const T *)a.ext_h + (active ? fi : 0) : nullptr;
    const T *yr = (const T *)a.y + tile * a.y_ts + lane;
    const T *yc = (const T *)a.y2 + tile * a.y2_ts + lane;
    T *es = FULL ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane : nullptr;
    T A[(NS + NM) * COLS];
    // bottom block first (Htilde, L, y die before the 156 state values are loaded: bounds the live set)
    {
        T Lw[tri(NM)], yv[NM];
#pragma unroll
        for (int e = 0; e < tri(NM); e++) Lw[e] = ldnt(mo, a.L.mo_LR + e);  // QUIRK srif.go:48: chol_L(R), not its inverse
#pragma unroll
        for (int r = 0; r < NM; r++) {
            const T re = active ? __builtin_nontemporal_load(yr + (int64_t)r * a.y_es) : T(0);
            const T co = active ? __builtin_nontemporal_load(yc + (int64_t)r * a.y2_es) : T(0);
            yv[r] = re - co;
            if constexpr (FULL) { if (active) stt(es, a.L.es_yhat + r, re); }
        }
#pragma unroll
        for (int r = 0; r < NM; r++)
#pragma unroll
            for (int j = 0; j < NS; j++) A[(NS + r) * COLS + j] = T(0);
#pragma unroll
        for (int l = 0; l < NM; l++) {
            T Hl[NS];
#pragma unroll
            for (int j = 0; j < NS; j++) Hl[j] = EXT ? __builtin_nontemporal_load(eh + (int64_t)(l * NS + j) * a.ext_ld) : ldnt(mo, a.L.mo_H + l * NS + j);
#pragma unroll
            for (int r = l; r < NM; r++)
#pragma unroll
                for (int j = 0; j < NS; j++) A[(NS + r) * COLS + j] += Lw[symi(l, r)] * Hl[j];  // (L Htilde)[r][j], l <= r
        }
#pragma unroll
        for (int r = 0; r < NM; r++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l <= r; l++) s += Lw[symi(l, r)] * yv[l];
            A[(NS + r) * COLS + NS] = s;
            if constexpr (FULL) { if (active) stt(es, a.L.es_dobs + r, s); }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NS; i++) {
#pragma unroll
        for (int j = 0; j < NS; j++) A[i * COLS + j] = ldt(st, NS + i * NS + j);
        A[i * COLS + NS] = ldt(st, i);
    }
    shouseholder<T, NS, NM>(A);
    T chk = T(0);
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = i; j < COLS; j++) chk += A[i * COLS + j] * T(0);
    const bool bad = chk != chk;   // stored as it is (helper.go:142-172 has no guard) and flagged
    if (active) {
#pragma unroll
        for (int i = 0; i < NS; i++) stt(st, i, A[i * COLS + NS]);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) stt(st, NS + i * NS + j, j >= i ? A[i * COLS + j] : T(0));
        if constexpr (FULL) {
#pragma unroll
            for (int r = 0; r < NM; r++) stt(es, a.L.es_innov + r, A[(NS + r) * COLS + NS]);
        }
    }
    if (active && bad) atomicOr(a.status + fi, (unsigned)KB_ST_NONFINITE);
}

// ---- fused Update for an upper-triangular R (the steady state) -----------------------------------
// R is upper triangular whenever its last writer was a measurement update (srif.go:334-337 zeroes the
// sub-columns) or the constructor (R0 diagonal, srif.go:20-29); the host tracks that (Batch::srif_tri, cleared by
// Predict(), which stores the full RBar).  Then the whole Update runs in one launch and RBar never goes to memory:
//   A  x = R^-1 b by back substitution; Phi -> VGPRs, xBar = Phi x; P Phi = L U in VGPRs; rows of RBar = R Phi^-1 two at
//      a time (the zeros of R's rows skipped), scattered with the row permutation undone into the LDS panel [RBar | bBar];
//   B  bottom block [L Htilde | L y] -> VGPRs, the panel LDS -> VGPRs, Householder, store b and the upper triangle of R.
// Per filter 339 values read, 90 written (1716 B in fp32) against the 2304 algorithmic bytes of the two-pass statement.
// One private [element][lane] LDS array per lane (conflict-free, no barriers: a lane only reads what it wrote).
// helper.go:142-172 HouseholderTransf like kb_static.h's shouseholder, calling row_done(k) as soon as row k is final
// (after step k nothing touches it again), so that finished rows leave the register file early.
template <typename T, int NN, int MM, typename F>
__device__ __forceinline__ void shouseholder_rows(T (&A)[(NN + MM) * (NN + 1)], F &&row_done) {
    constexpr int ROWS = NN + MM, COLS = NN + 1;
#pragma unroll
    for (int k = 0; k < NN; k++) {
        T sigma = T(0);
#pragma unroll
        for (int i = k; i < ROWS; i++) sigma += A[i * COLS + k] * A[i * COLS + k];
        const T akk = A[k * COLS + k];
        const T sgn = (akk == T(0) || fabs(akk) <= T(1e-12)) ? T(1) : copysign(T(1), akk);  // helper.go:133-138 Sign
        sigma = sqrt(sigma) * sgn;
        const T uk = akk + sigma;
        A[k * COLS + k] = -sigma;
        const T beta = T(1) / (sigma * uk);
#pragma unroll
        for (int j = k + 1; j < COLS; j++) {
            T gamma = uk * A[k * COLS + j];
#pragma unroll
            for (int i = k + 1; i < ROWS; i++) gamma += A[i * COLS + k] * A[i * COLS + j];
            gamma *= beta;
            A[k * COLS + j] = A[k * COLS + j] - gamma * uk;
#pragma unroll
            for (int i = k + 1; i < ROWS; i++) A[i * COLS + j] = A[i * COLS + j] - gamma * A[i * COLS + k];
        }
        row_done(k);
    }
}

template <typename T, int NS>
constexpr bool srif_fused_fits() { return sizeof(T) * 4 * (NS * (NS + 1) + 1) * KB_TILE + 64 <= 160 * 1024; }

// Phase A for one tile: (b, R upper, Phi) -> the LDS panel [RBar | bBar] plus an "ok" slot (0 when Phi / R turned out
// singular at this step: the filter keeps its estimate for this step; the status word is updated here).
// TRI = false is the cold variant for a tile in which some filter may hold a dense R although the batch as a whole is in
// the triangular steady state: a filter that skipped the Update right after a Predict() (singular Phi) still has the full
// RBar of that Predict().  Such filters carry a non-zero status word, so the kernel picks the variant per tile from the
// status words (wave-uniform); the dense variant reads all of R and solves State(prev) by pivoted LU.
template <typename T, int NS, bool EXT, bool TRI>
__device__ __forceinline__ void srif_time_to_panel(const StepArgs &a, int64_t tile, int lane, T *panel) {
    constexpr int COLS = NS + 1, RG = 2;
    static_assert(NS % RG == 0, "row groups");
    const int64_t fi = tile * KB_TILE + lane;
    const bool inb = fi < a.N;
    const bool active = inb;
    const T *st = (const T *)a.state + tile * ((int64_t)KB_TILE * (NS + NS * NS)) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *ephi = EXT ? (const T *)a.ext_phi + (inb ? fi : 0) : nullptr;
    unsigned err = 0;
    T xprev[NS], lu[NS * NS];
    if constexpr (!TRI) {
        T Rw[NS * NS];
#pragma unroll
        for (int i = 0; i < NS; i++) xprev[i] = ldt(st, i);
#pragma unroll
        for (int e = 0; e < NS * NS; e++) Rw[e] = ldt(st, NS + e);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) panel[(i * COLS + j) * KB_TILE] = Rw[i * NS + j];
        if (lu_solve_inplace<T, NS, 1>(Rw, xprev)) err |= KB_ST_SINGULAR;   // State(prev) = R^-1 b (srif.go:223-234)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < NS * NS; e++) lu[e] = EXT ? __builtin_nontemporal_load(ephi + (int64_t)e * a.ext_ld) : ldnt(mo, a.L.mo_F + e);
    } else {
        T Ru[tri(NS)];   // upper triangle of R, Ru[symi(i, j)], i <= j
#pragma unroll
        for (int i = 0; i < NS; i++) xprev[i] = ldt(st, i);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = i; j < NS; j++) Ru[symi(i, j)] = ldt(st, NS + i * NS + j);
#pragma unroll
        for (int e = 0; e < NS * NS; e++) lu[e] = EXT ? __builtin_nontemporal_load(ephi + (int64_t)e * a.ext_ld) : ldnt(mo, a.L.mo_F + e);
        __builtin_amdgcn_sched_barrier(0);
        // State(prev) = R^-1 b by back substitution (srif.go:223-234); the rows of R wait in the LDS panel for the solves
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = i; j < NS; j++) panel[(i * COLS + j) * KB_TILE] = Ru[symi(i, j)];
#pragma unroll
        for (int i = NS - 1; i >= 0; i--) {
            T sum = xprev[i];
#pragma unroll
            for (int k2 = i + 1; k2 < NS; k2++) sum -= Ru[symi(i, k2)] * xprev[k2];
            if (Ru[symi(i, i)] == T(0)) err |= KB_ST_SINGULAR;
            xprev[i] = sum * (T(1) / Ru[symi(i, i)]);
        }
    }
#pragma unroll
    for (int i = 0; i < NS; i++) {   // xBar = Phi State(prev) (srif.go:118) -> LDS, read back in pivoted order below
        T s = T(0);
#pragma unroll
        for (int j = 0; j < NS; j++) s += lu[i * NS + j] * xprev[j];
        panel[(i * COLS + NS) * KB_TILE] = s;   // the bBar slots are free until the solves
    }
    // P Phi = L U (srif.go:111-114's Inverse = Dgetrf + ...): nibble k of perm = original index of the row now in position k
    uint64_t perm = 0xBA9876543210ull;
#pragma unroll
    for (int j = 0; j < NS; j++) {
#pragma unroll
        for (int r = j + 1; r < NS; r++) {
            const bool sw = fabs(lu[r * NS + j]) > fabs(lu[j * NS + j]);
            if (__any(sw)) {
#pragma unroll
                for (int c = 0; c < NS; c++) {
                    const T t0 = lu[j * NS + c], t1 = lu[r * NS + c];
                    lu[j * NS + c] = sw ? t1 : t0;
                    lu[r * NS + c] = sw ? t0 : t1;
                }
                const uint64_t x = sw ? (((perm >> (4 * j)) ^ (perm >> (4 * r))) & 15u) : 0u;
                perm ^= (x << (4 * j)) | (x << (4 * r));
            }
        }
        const T piv = lu[j * NS + j];
        if (piv == T(0)) err |= KB_ST_SINGULAR;
        const T rp = T(1) / piv;
        lu[j * NS + j] = rp;   // the solves multiply by the reciprocal
#pragma unroll
        for (int r = j + 1; r < NS; r++) {
            const T l = lu[r * NS + j] * rp;
            lu[r * NS + j] = l;
#pragma unroll
            for (int c = j + 1; c < NS; c++) lu[r * NS + c] -= l * lu[j * NS + c];
        }
    }
    if (err && active) atomicOr(a.status + fi, err);
    panel[NS * COLS * KB_TILE] = (active && !err) ? T(1) : T(0);   // a failed filter keeps its estimate: the measurement half skips it
    int poff[NS];
    T xBarP[NS];
#pragma unroll
    for (int r = 0; r < NS; r++) {
        poff[r] = nib(perm, r) * KB_TILE;
        xBarP[r] = panel[nib(perm, r) * (COLS * KB_TILE) + NS * KB_TILE];
    }
    // RBar = R Phi^-1 (srif.go:115): row i solves z Phi = R[i,:], i.e. w U = r, v L = w, z[perm_k] = v_k; R[i, c] = 0 for c < i
#pragma unroll
    for (int i0 = 0; i0 < NS; i0 += RG) {
        T z[RG][NS];
#pragma unroll
        for (int g = 0; g < RG; g++)
#pragma unroll
            for (int c = 0; c < NS; c++) z[g][c] = (!TRI || c >= i0 + g) ? panel[((i0 + g) * COLS + c) * KB_TILE] : T(0);
#pragma unroll
        for (int j = 0; j < NS; j++) {      // constant trip counts everywhere: the structural-zero tests fold after unrolling
#pragma unroll
            for (int g = 0; g < RG; g++) {
                if (!TRI || j >= i0 + g) {
                    T sum = z[g][j];
#pragma unroll
                    for (int k2 = 0; k2 < NS; k2++)
                        if ((!TRI || k2 >= i0 + g) && k2 < j) sum -= z[g][k2] * lu[k2 * NS + j];
                    z[g][j] = sum * lu[j * NS + j];
                }
            }
        }
#pragma unroll
        for (int j = NS - 2; j >= 0; j--) {
#pragma unroll
            for (int g = 0; g < RG; g++) {
                T sum = z[g][j];
#pragma unroll
                for (int k2 = j + 1; k2 < NS; k2++) sum -= z[g][k2] * lu[k2 * NS + j];
                z[g][j] = sum;
            }
        }
#pragma unroll
        for (int g = 0; g < RG; g++) {
            T bb = T(0);
#pragma unroll
            for (int r = 0; r < NS; r++) {
                bb += z[g][r] * xBarP[r];                                   // :119 bBar = RBar xBar (pivoted order)
                panel[poff[r] + (i0 + g) * COLS * KB_TILE] = z[g][r];       // row permutation undone by the scatter
            }
            panel[((i0 + g) * COLS + NS) * KB_TILE] = bb;
        }
    }
}

// Phase B for one tile: the LDS panel [RBar | bBar] + (Htilde, chol_L(R), real, computed) -> Householder -> b, R (upper).
template <typename T, int NS, int NM, bool FULL, bool EXT, typename W, typename F>
__device__ __forceinline__ void srif_meas_from_panel(const StepArgs &a, int64_t tile, int lane, const T *panel, bool dense, W &&wait_panel, F &&panel_consumed) {
    constexpr int COLS = NS + 1;
    const int64_t fi = tile * KB_TILE + lane;
    const bool inb = fi < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (NS + NS * NS)) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *eh = EXT ? (const T *)a.ext_h + (inb ? fi : 0) : nullptr;
    const T *yr = (const T *)a.y + tile * a.y_ts + lane;
    const T *yc = (const T *)a.y2 + tile * a.y2_ts + lane;
    T *es = FULL ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane : nullptr;
    T Hh[NM * NS], Lw[tri(NM)], yv[NM];
    [[maybe_unused]] T yreal[NM];
#pragma unroll
    for (int e = 0; e < NM * NS; e++) Hh[e] = EXT ? __builtin_nontemporal_load(eh + (int64_t)e * a.ext_ld) : ldnt(mo, a.L.mo_H + e);
#pragma unroll
    for (int e = 0; e < tri(NM); e++) Lw[e] = ldnt(mo, a.L.mo_LR + e);  // QUIRK srif.go:48: chol_L(R), not its inverse
#pragma unroll
    for (int r = 0; r < NM; r++) {
        const T re = inb ? __builtin_nontemporal_load(yr + (int64_t)r * a.y_es) : T(0);
        const T co = inb ? __builtin_nontemporal_load(yc + (int64_t)r * a.y2_es) : T(0);
        yv[r] = re - co;
        if constexpr (FULL) yreal[r] = re;
    }
    wait_panel();   // the loads above are in flight while phase A of this tile finishes
    const bool active = inb && panel[NS * COLS * KB_TILE] != T(0);
    // ---- measurement update (srif.go:143-156, :298-340): Householder on [[RBar bBar],[L Htilde, L y]]
    T A[(NS + NM) * COLS];
    {
        if constexpr (FULL) {
            if (active) {
#pragma unroll
                for (int r = 0; r < NM; r++) stt(es, a.L.es_yhat + r, yreal[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < NM; r++)
#pragma unroll
            for (int j = 0; j < NS; j++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l <= r; l++) s += Lw[symi(l, r)] * Hh[l * NS + j];  // (L Htilde)[r][j], l <= r
                A[(NS + r) * COLS + j] = s;
            }
#pragma unroll
        for (int r = 0; r < NM; r++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l <= r; l++) s += Lw[symi(l, r)] * yv[l];
            A[(NS + r) * COLS + NS] = s;
            if constexpr (FULL) { if (active) stt(es, a.L.es_dobs + r, s); }
        }
    }
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < COLS; j++) {
            A[i * COLS + j] = panel[(i * COLS + j) * KB_TILE];
            if constexpr (FULL) { if (active && j < NS) stt(es, a.L.es_ppred + i * NS + j, A[i * COLS + j]); }
        }
    panel_consumed();   // the buffer may be refilled while the Householder runs on registers
    // A non-finite result is stored as it is (what the reference's unguarded HouseholderTransf leaves behind,
    // helper.go:142-172) and flagged in the status word.
    T chk = T(0);
    shouseholder_rows<T, NS, NM>(A, [&](int k) {
#pragma unroll
        for (int j = 0; j < COLS; j++)
            if (j >= k) chk += A[k * COLS + j] * T(0);
        if (active) {
            stt(st, k, A[k * COLS + NS]);
#pragma unroll
            for (int j = 0; j < NS; j++)
                if (j >= k) stt(st, NS + k * NS + j, A[k * COLS + j]);   // the lower triangle holds zeros already
        }
    });
    if constexpr (FULL) {
        if (active) {
#pragma unroll
            for (int r = 0; r < NM; r++) stt(es, a.L.es_innov + r, A[(NS + r) * COLS + NS]);
        }
    }
    if (dense && active) {   // cold: this filter's R may have been the dense RBar of a Predict() (see srif_time_to_panel<..., TRI = false>)
#pragma unroll
        for (int i = 1; i < NS; i++)
#pragma unroll
            for (int j = 0; j < i; j++) stt(st, NS + i * NS + j, T(0));   // srif.go:334-337 zeroes the sub-columns
    }
    if (active && chk != chk) atomicOr(a.status + fi, (unsigned)KB_ST_NONFINITE);
}

// The fused Update as a two-stage pipeline inside a workgroup of two waves: wave 0 runs phase A (time update) of the
// workgroup's tiles into two LDS panel buffers, wave 1 runs phase B (measurement update) out of them.  The waves hand
// buffers over through LDS flags, not barriers, so that they drift apart: one wave's loads are in flight while its
// partner computes, which a single wave doing both halves cannot arrange (no registers left to prefetch into) and a
// barrier per tile would undo (both would load, then both compute).  Persistent grid (two workgroups per CU:
// 2 x 2 x 39 KB of LDS), tiles strided by the grid size.
template <typename T, int NS, int NM, bool FULL, bool EXT>
__global__ void __launch_bounds__(128, 1) srif_fused_kernel(const StepArgs a) {
    constexpr int SLOTS = NS * (NS + 1) + 1;
    __shared__ T lds[srif_fused_fits<T, NS>() ? 2 * SLOTS * KB_TILE : 1];
    __shared__ int full[2];   // 1: the panel buffer holds a tile for wave 1; 0: wave 0 may (re)fill it
    const int lane = threadIdx.x & 63;
    const bool first_half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;
    if (threadIdx.x < 2) full[threadIdx.x] = 0;
    __syncthreads();
    const int64_t stride = gridDim.x;
    int it = 0;
    for (int64_t cur = blockIdx.x; cur < a.ntiles; cur += stride, it++) {
        const int bsel = it & 1;
        T *panel = lds + bsel * (SLOTS * KB_TILE) + lane;
        auto wait_for = [&](int want) {
            while (__hip_atomic_load(&full[bsel], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != want) __builtin_amdgcn_s_sleep(8);
        };
        // a non-zero status word in the tile: some filter may hold a dense R (it skipped the Update after a Predict())
        const int64_t fi = cur * KB_TILE + lane;
        const bool dense = __any(fi < a.N && (a.status[fi < a.N ? fi : 0] & ~KB_ST_SKIP_STEP) != 0u);
        if (first_half) {
            wait_for(0);
            if (dense) srif_time_to_panel<T, NS, EXT, false>(a, cur, lane, panel);
            else srif_time_to_panel<T, NS, EXT, true>(a, cur, lane, panel);
            __hip_atomic_store(&full[bsel], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            srif_meas_from_panel<T, NS, NM, FULL, EXT>(a, cur, lane, panel, dense, [&]() { wait_for(1); }, [&]() {
                __hip_atomic_store(&full[bsel], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            });
        }
    }
}

static int num_cus(int device) {
    static int cached[64] = {0};
    if (device < 0 || device >= 64) return 256;
    if (!cached[device]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
        cached[device] = n;
    }
    return cached[device];
}

static bool srif_shape_ok(const StepArgs &a, int NS, int NM) { return a.n == NS && a.p == NM; }

template <typename T, int NS, int NM>
static bool srif_try(const Batch &b, const StepArgs &a) {
    if (!srif_shape_ok(a, NS, NM)) return false;
    const dim3 grid = tile_grid(a.ntiles), block(256);
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0, ext = a.ext_phi != nullptr;
    if constexpr (srif_fused_fits<T, NS>()) {
        if (!a.predict && a.srif_tri) {
            const int64_t slots = 2 * (int64_t)num_cus(b.device);   // two resident workgroups per CU
#define KB_F(F_, E_) hipLaunchKernelGGL((srif_fused_kernel<T, NS, NM, F_, E_>), dim3((unsigned)(a.ntiles < slots ? a.ntiles : slots)), dim3(128), 0, b.stream, a)
            if (full) { if (ext) KB_F(true, true); else KB_F(true, false); }
            else      { if (ext) KB_F(false, true); else KB_F(false, false); }
#undef KB_F
            return true;
        }
    }
#define KB_T(F_, E_) hipLaunchKernelGGL((srif_time_kernel<T, NS, F_, E_>), grid, block, 0, b.stream, a)
#define KB_M(F_, E_) hipLaunchKernelGGL((srif_meas_kernel<T, NS, NM, F_, E_>), grid, block, 0, b.stream, a)
    if (full) { if (ext) KB_T(true, true); else KB_T(true, false); }
    else      { if (ext) KB_T(false, true); else KB_T(false, false); }
    if (!a.predict) {
        if (full) { if (ext) KB_M(true, true); else KB_M(true, false); }
        else      { if (ext) KB_M(false, true); else KB_M(false, false); }
    }
#undef KB_T
#undef KB_M
    return true;
}

bool srif_reg_ok(const Batch &b, const StepArgs &a) {
    if (b.dtype == KB_F32) return srif_shape_ok(a, 12, 6) || srif_shape_ok(a, 6, 2);
    return srif_shape_ok(a, 6, 2) || srif_shape_ok(a, 12, 6);
}

int launch_srif(const Batch &b, const StepArgs &a) {
    bool done = false;
    // Update(): the two-lanes-per-filter kernel (kb_srif_pair.h); Predict() and KB_SRIF_ONE_LANE=1 (comparison runs): the
    // one-filter-per-lane kernels of this file
    static const bool one_lane = getenv("KB_SRIF_ONE_LANE") != nullptr;
    if (!a.predict && !one_lane) {
        done = b.dtype == KB_F32 ? launch_srif_pair_f32(b, a) : launch_srif_pair_f64(b, a);
        if (done) { KB_HIP(hipGetLastError()); return KB_OK; }
    }
    if (b.dtype == KB_F32) done = srif_try<float, 12, 6>(b, a) || srif_try<float, 6, 2>(b, a);
    else done = srif_try<double, 6, 2>(b, a) || srif_try<double, 12, 6>(b, a);
    if (!done) return launch_srif_gen(b, a);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
