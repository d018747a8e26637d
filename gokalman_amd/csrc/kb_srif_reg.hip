// kb_srif_reg.hip -- register-resident SRIF update (srif.go:101-160, :298-340, helper.go:142-172)
// for the benchmark shape (n = 12, p = 6, fp32) and the reference tests' shape (n = 6, p = 2).
// One filter per lane; the 18 x 13 Householder panel, Phi^-1 and the LU work arrays live in
// VGPRs (1 wave/SIMD, 512-register budget); Phi / H-tilde are read in place from the caller's
// planar arrays after kb_prepare_dev (zero-copy) or from the model block after kb_prepare.
//
// Two launches per Update, mirroring the reference's own split: the time update
// (srif.go:111-141, also the whole of Predict()) rewrites (b, R) as (bBar, RBar) in place, the
// measurement update (srif.go:143-156, :298-340) runs the Householder panel on it.  RBar
// (164 MB at 256k filters) stays in the Infinity Cache between the two.
// Differences from the statement-by-statement generic kernel (rounding level only):
//   State(prev) = R^-1 b and RBar = R Phi^-1 are obtained by LU solves instead of
//   inverse-then-multiply (srif.go:111-115, :223-234); only exact singularity / non-finite
//   results are flagged (the generic kernel also applies gonum's cond > 1e16 test); a filter
//   whose status word is non-zero is skipped by the measurement kernel until kb_clear_status.
// Algorithmic bytes per filter-step: b 12 + R 144 + Phi 144 + Htilde 72 +
// L 36 + real 6 + computed 6 read, b 12 + R 144 written = 576 elements = 2304 B in fp32.
#include "kb_internal.h"
#include "kb_static.h"

namespace kb {


// LU factorisation with partial pivoting in place (unit-lower L below the diagonal, U on and
// above), recording every row exchange as one bit (exchange index = position in the (j, r) loop
// nest) so that later right-hand sides can be permuted without keeping a permutation matrix.
template <typename T, int P>
__device__ __forceinline__ bool lu_factor_record(T (&a)[P * P], unsigned (&bits)[(P * (P - 1) / 2 + 31) / 32]) {
    bool bad = false;
#pragma unroll
    for (int w = 0; w < (P * (P - 1) / 2 + 31) / 32; w++) bits[w] = 0u;
    int idx = 0;
#pragma unroll
    for (int j = 0; j < P; j++) {
#pragma unroll
        for (int r = j + 1; r < P; r++) {
            const bool sw = fabs(a[r * P + j]) > fabs(a[j * P + j]);
            bits[idx >> 5] |= (sw ? 1u : 0u) << (idx & 31);
            idx++;
#pragma unroll
            for (int c = 0; c < P; c++) {  // whole rows: the L part moves with its row (LAPACK dlaswp)
                const T t0 = a[j * P + c], t1 = a[r * P + c];
                a[j * P + c] = sw ? t1 : t0;
                a[r * P + c] = sw ? t0 : t1;
            }
        }
        const T piv = a[j * P + j];
        bad = bad || (piv == T(0));
        const T rp = T(1) / piv;
#pragma unroll
        for (int r = j + 1; r < P; r++) {
            const T l = a[r * P + j] * rp;
            a[r * P + j] = l;
#pragma unroll
            for (int c = j + 1; c < P; c++) a[r * P + c] -= l * a[j * P + c];
        }
    }
    return bad;
}

// z = r A^-1 for a row vector r, given the recorded LU of A (P A = L U): w U = r, v L = w, z = v P.
template <typename T, int P>
__device__ __forceinline__ void lu_row_solve(const T (&lu)[P * P], const unsigned (&bits)[(P * (P - 1) / 2 + 31) / 32], T (&z)[P]) {
#pragma unroll
    for (int j = 0; j < P; j++) {  // w U = r
        T s = z[j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= z[k] * lu[k * P + j];
        z[j] = s / lu[j * P + j];
    }
#pragma unroll
    for (int j = P - 1; j >= 0; j--) {  // v L = w (unit lower)
        T s = z[j];
#pragma unroll
        for (int k = j + 1; k < P; k++) s -= z[k] * lu[k * P + j];
        z[j] = s;
    }
    int idx = P * (P - 1) / 2 - 1;  // z = v P: undo the exchanges in reverse order
#pragma unroll
    for (int j = P - 1; j >= 0; j--)
#pragma unroll
        for (int r = P - 1; r > j; r--) {
            const bool sw = (bits[idx >> 5] >> (idx & 31)) & 1u;
            idx--;
            const T t0 = z[j], t1 = z[r];
            z[j] = sw ? t1 : t0;
            z[r] = sw ? t0 : t1;
        }
}

// ---- time update (srif.go:111-141): b <- bBar, R <- RBar = R Phi^-1, in place -------------------
template <typename T, int NS, bool FULL, bool EXT>
__global__ void __launch_bounds__(256, 1) srif_time_kernel(const StepArgs a) {
    constexpr int NB = (NS * (NS - 1) / 2 + 31) / 32;
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (NS + NS * NS)) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *ephi = EXT ? (const T *)a.ext_phi + (active ? fi : 0) : nullptr;
    unsigned err = 0;
    T xprev[NS];
    {   // State(prev) = R^-1 b (srif.go:223-234)
        T Rw[NS * NS];
#pragma unroll
        for (int i = 0; i < NS; i++) xprev[i] = ldt(st, i);
#pragma unroll
        for (int e = 0; e < NS * NS; e++) Rw[e] = ldt(st, NS + e);
        if (lu_solve_inplace<T, NS, 1>(Rw, xprev)) err |= KB_ST_SINGULAR;
    }
    // keep the machine scheduler from hoisting the next phase's loads above this one: the phases are
    // sized to fit the register file one at a time (other waves cover the load latency)
    __builtin_amdgcn_sched_barrier(0);
    T Phi[NS * NS], xBar[NS];
    unsigned bits[NB];
#pragma unroll
    for (int e = 0; e < NS * NS; e++) Phi[e] = EXT ? __builtin_nontemporal_load(ephi + (int64_t)e * a.ext_ld) : ldnt(mo, a.L.mo_F + e);
    smv<T, NS, NS>(Phi, xprev, xBar);                      // :118 xBar = Phi State(prev)
    if (lu_factor_record<T, NS>(Phi, bits)) err |= KB_ST_SINGULAR;  // :111-114
    if (err) { if (active) atomicOr(a.status + fi, err); return; }
    T *es = FULL ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane : nullptr;
    T bBar[NS], znext[NS];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int l = 0; l < NS; l++) znext[l] = ldt(st, NS + l);  // row 0 of R again: cache hit
#pragma unroll
    for (int i = 0; i < NS; i++) {
        T z[NS];
#pragma unroll
        for (int l = 0; l < NS; l++) z[l] = znext[l];
        if (i + 1 < NS) {
#pragma unroll
            for (int l = 0; l < NS; l++) znext[l] = ldt(st, NS + (i + 1) * NS + l);  // prefetch the next row
        }
        __builtin_amdgcn_sched_barrier(0);
        lu_row_solve<T, NS>(Phi, bits, z);                 // :115 row i of RBar = R Phi^-1
        T bb = T(0);
#pragma unroll
        for (int j = 0; j < NS; j++) bb += z[j] * xBar[j];  // :119 bBar = RBar xBar
        bBar[i] = bb;
        if (active) {
#pragma unroll
            for (int j = 0; j < NS; j++) {
                stt(st, NS + i * NS + j, z[j]);
                if constexpr (FULL) stt(es, a.L.es_ppred + i * NS + j, z[j]);
            }
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < NS; i++) stt(st, i, bBar[i]);
    }
}

// ---- time update, LDS-resident LU (n = 12 fp32) ------------------------------------------------
// The 12 x 12 pivoted LU of Phi does not fit the register file once unrolled (see the header of
// this file), so this variant keeps Phi / its LU factors in LDS, one private 144-element array per
// lane laid out [element][lane]: lane l always hits bank l, whatever element it indexes, so
// per-lane *dynamic* row indices are conflict-free and partial pivoting becomes a per-lane row
// permutation (12 nibbles in a 64-bit register) instead of data movement.  One workgroup of
// 4 waves per CU (4 x 39 KB of LDS).
template <typename T, int NS>
struct LdsLU {
    T *base;  // this lane's element 0; element e at base[e * 64]
    __device__ __forceinline__ T get(int row, int col) const { return base[(row * NS + col) * KB_TILE]; }
    __device__ __forceinline__ void put(int row, int col, T v) const { base[(row * NS + col) * KB_TILE] = v; }
};
__device__ __forceinline__ int nib(uint64_t perm, int r) { return (int)((perm >> (4 * r)) & 15u); }

template <typename T, int NS, bool FULL, bool EXT>
__global__ void __launch_bounds__(256, 1) srif_time_lds_kernel(const StepArgs a) {
    __shared__ T lds[4 * (NS * NS + NS) * KB_TILE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (NS + NS * NS)) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *ephi = EXT ? (const T *)a.ext_phi + (active ? fi : 0) : nullptr;
    const LdsLU<T, NS> lu{lds + wv * (NS * NS + NS) * KB_TILE + lane};
    T *ltmp = lds + wv * (NS * NS + NS) * KB_TILE + NS * NS * KB_TILE + lane;  // NS spare elements per lane
    unsigned err = 0;
    T xprev[NS];
    {   // State(prev) = R^-1 b (srif.go:223-234), register LU solve
        T Rw[NS * NS];
#pragma unroll
        for (int i = 0; i < NS; i++) xprev[i] = ldt(st, i);
#pragma unroll
        for (int e = 0; e < NS * NS; e++) Rw[e] = ldt(st, NS + e);
        if (lu_solve_inplace<T, NS, 1>(Rw, xprev)) err |= KB_ST_SINGULAR;
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // Phi -> LDS; xBar = Phi State(prev) (srif.go:118) -> spare slots
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NS; j++) {
                const T v = EXT ? __builtin_nontemporal_load(ephi + (int64_t)(i * NS + j) * a.ext_ld) : ldnt(mo, a.L.mo_F + i * NS + j);
                lu.put(i, j, v);
                s += v * xprev[j];
            }
            ltmp[i * KB_TILE] = s;
        }
    }
    // LU with partial pivoting (srif.go:111-114's Inverse = Dgetrf + ...): logical row r lives in
    // physical row nib(perm, r)
    uint64_t perm = 0xBA9876543210ull;
#pragma unroll
    for (int j = 0; j < NS; j++) {
        T best = T(-1);
        int p = j;
#pragma unroll
        for (int r = j; r < NS; r++) {  // all reads issue back to back (static nibble positions), then a select chain
            const T v = fabs(lu.get(nib(perm, r), j));
            const bool g = v > best;
            best = g ? v : best;
            p = g ? r : p;
        }
        const uint64_t vj = (perm >> (4 * j)) & 15u, vp = (perm >> (4 * p)) & 15u, x = vj ^ vp;
        perm ^= (x << (4 * j)) | (x << (4 * p));
        const int pj = nib(perm, j);
        T prow[NS];
#pragma unroll
        for (int c = j; c < NS; c++) prow[c] = lu.get(pj, c);
        if (prow[j] == T(0)) err |= KB_ST_SINGULAR;
        const T rp = T(1) / prow[j];
#pragma unroll
        for (int r = j + 1; r < NS; r++) {
            const int pr = nib(perm, r);
            const T l = lu.get(pr, j) * rp;
            lu.put(pr, j, l);
#pragma unroll
            for (int c = j + 1; c < NS; c++) lu.put(pr, c, lu.get(pr, c) - l * prow[c]);
        }
    }
    if (err) { if (active) atomicOr(a.status + fi, err); return; }
    T xBarP[NS];  // xBar in pivoted order: xBarP[r] = xBar[perm_r]
#pragma unroll
    for (int r = 0; r < NS; r++) xBarP[r] = ltmp[nib(perm, r) * KB_TILE];
    T *es = FULL ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane : nullptr;
    // RBar = R Phi^-1 row by row, RG rows at a time so that every LU element read from LDS serves RG
    // right-hand sides; the next group's rows of R are prefetched while the current group is solved.
    constexpr int RG = (NS % 4 == 0) ? 4 : ((NS % 3 == 0) ? 3 : 1);
    int prow[NS];  // physical row of logical row r
#pragma unroll
    for (int r = 0; r < NS; r++) prow[r] = nib(perm, r);
    T znext[RG][NS];
#pragma unroll
    for (int g = 0; g < RG; g++)
#pragma unroll
        for (int l = 0; l < NS; l++) znext[g][l] = ldt(st, NS + g * NS + l);  // rows of R again: cache hits
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
        // z Phi = r  with  P Phi = L U:  w U = r,  v L = w,  z[perm_r] = v_r   (srif.go:115)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            T s[RG];
#pragma unroll
            for (int g = 0; g < RG; g++) s[g] = z[g][j];
#pragma unroll
            for (int k2 = 0; k2 < j; k2++) {
                const T u = lu.get(prow[k2], j);
#pragma unroll
                for (int g = 0; g < RG; g++) s[g] -= z[g][k2] * u;
            }
            const T d = lu.get(prow[j], j);
#pragma unroll
            for (int g = 0; g < RG; g++) z[g][j] = s[g] / d;
        }
#pragma unroll
        for (int j = NS - 1; j >= 0; j--) {
            T s[RG];
#pragma unroll
            for (int g = 0; g < RG; g++) s[g] = z[g][j];
#pragma unroll
            for (int k2 = j + 1; k2 < NS; k2++) {
                const T l = lu.get(prow[k2], j);
#pragma unroll
                for (int g = 0; g < RG; g++) s[g] -= z[g][k2] * l;
            }
#pragma unroll
            for (int g = 0; g < RG; g++) z[g][j] = s[g];
        }
#pragma unroll
        for (int g = 0; g < RG; g++) {
            T bb = T(0);
#pragma unroll
            for (int r = 0; r < NS; r++) {
                bb += z[g][r] * xBarP[r];             // :119 bBar = RBar xBar (same products, pivoted order)
                ltmp[prow[r] * KB_TILE] = z[g][r];    // un-permute through the spare slots
            }
            if (active) {
#pragma unroll
                for (int c = 0; c < NS; c++) {
                    const T v = ltmp[c * KB_TILE];
                    stt(st, NS + (i0 + g) * NS + c, v);
                    if constexpr (FULL) stt(es, a.L.es_ppred + (i0 + g) * NS + c, v);
                }
                stt(st, i0 + g, bb);  // b <- bBar: rows i0.. of R and their b entries are not read again
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
    if (active && (a.status[fi] & (KB_ST_SINGULAR | KB_ST_ASYMMETRIC | KB_ST_NONFINITE)) != 0u) return;  // failed (now or earlier): the estimate stays frozen
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
    const bool bad = chk != chk;
    if (active && !bad) {
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

static bool srif_shape_ok(const StepArgs &a, int NS, int NM) { return a.n == NS && a.p == NM; }

template <typename T, int NS, int NM>
static bool srif_try(const Batch &b, const StepArgs &a) {
    if (!srif_shape_ok(a, NS, NM)) return false;
    const dim3 grid = tile_grid(a.ntiles), block(256);
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0, ext = a.ext_phi != nullptr;
#define KB_T(F_, E_) do { if constexpr (NS == 12 && sizeof(T) == 4) hipLaunchKernelGGL((srif_time_lds_kernel<T, NS, F_, E_>), grid, block, 0, b.stream, a); \
                            else hipLaunchKernelGGL((srif_time_kernel<T, NS, F_, E_>), grid, block, 0, b.stream, a); } while (0)
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
    return srif_shape_ok(a, 6, 2);
}

int launch_srif(const Batch &b, const StepArgs &a) {
    bool done = false;
    if (b.dtype == KB_F32) done = srif_try<float, 12, 6>(b, a) || srif_try<float, 6, 2>(b, a);
    else done = srif_try<double, 6, 2>(b, a);
    if (!done) return launch_srif_gen(b, a);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
