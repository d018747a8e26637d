// kb_srif_reg.hip -- register-resident SRIF update (srif.go:101-160, :298-340, helper.go:142-172)
// for the benchmark shape (n = 12, p = 6, fp32) and the reference tests' shape (n = 6, p = 2).
// One filter per lane; the 18 x 13 Householder panel, Phi^-1 and the LU work arrays live in
// VGPRs (1 wave/SIMD, 512-register budget); Phi / H-tilde are read in place from the caller's
// planar arrays after kb_prepare_dev (zero-copy) or from the model block after kb_prepare.
//
// Differences from the statement-by-statement generic kernel (rounding level only):
//   State(prev) = R^-1 b is obtained by an LU solve instead of inverse-then-multiply
//   (srif.go:223-234), and only exact singularity / non-finite results are flagged (the
//   generic kernel also applies gonum's cond > 1e16 test).
// Algorithmic bytes per filter-step: b 12 + R 144 + Phi 144 + Htilde 72 +
// L 36 + real 6 + computed 6 read, b 12 + R 144 written = 576 elements = 2304 B in fp32.
#include "kb_internal.h"
#include "kb_static.h"

namespace kb {

template <typename T>
__device__ __forceinline__ T sl(const T *p, int e) { return p[(int64_t)e * KB_TILE]; }
template <typename T>
__device__ __forceinline__ T snt(const T *p, int e) { return __builtin_nontemporal_load(p + (int64_t)e * KB_TILE); }
template <typename T>
__device__ __forceinline__ void ss(T *p, int e, T v) { p[(int64_t)e * KB_TILE] = v; }

// Gaussian elimination with partial pivoting on registers, right-hand sides solved in place.
// a[P*P] is destroyed, b[P*C] becomes a^-1 b.  Returns true on an exact zero pivot.
template <typename T, int P, int C>
__device__ __forceinline__ bool lu_solve_inplace(T (&a)[P * P], T (&b)[P * C]) {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < P; j++) {
#pragma unroll
        for (int r = j + 1; r < P; r++) {
            const bool sw = fabs(a[r * P + j]) > fabs(a[j * P + j]);
#pragma unroll
            for (int c = j; c < P; c++) {
                const T t0 = a[j * P + c], t1 = a[r * P + c];
                a[j * P + c] = sw ? t1 : t0;
                a[r * P + c] = sw ? t0 : t1;
            }
#pragma unroll
            for (int c = 0; c < C; c++) {
                const T u0 = b[j * C + c], u1 = b[r * C + c];
                b[j * C + c] = sw ? u1 : u0;
                b[r * C + c] = sw ? u0 : u1;
            }
        }
        const T piv = a[j * P + j];
        bad = bad || (piv == T(0));
        const T rp = T(1) / piv;
#pragma unroll
        for (int r = j + 1; r < P; r++) {
            const T l = a[r * P + j] * rp;
#pragma unroll
            for (int c = j + 1; c < P; c++) a[r * P + c] -= l * a[j * P + c];
#pragma unroll
            for (int c = 0; c < C; c++) b[r * C + c] -= l * b[j * C + c];
        }
    }
#pragma unroll
    for (int i = P - 1; i >= 0; i--) {
        const T rd = T(1) / a[i * P + i];
#pragma unroll
        for (int c = 0; c < C; c++) {
            T s = b[i * C + c];
#pragma unroll
            for (int k = i + 1; k < P; k++) s -= a[i * P + k] * b[k * C + c];
            b[i * C + c] = s * rd;
        }
    }
    return bad;
}

template <typename T, int NS, int NM, bool FULL, bool EXT>
__global__ void __launch_bounds__(256, 1) srif_reg_kernel(const StepArgs a) {
    constexpr int COLS = NS + 1;
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (NS + NS * NS)) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *ephi = EXT ? (const T *)a.ext_phi + (active ? fi : 0) : nullptr;
    const T *eh = EXT ? (const T *)a.ext_h + (active ? fi : 0) : nullptr;
    const T *yr = (const T *)a.y + tile * a.y_ts + lane;
    const T *yc = (const T *)a.y2 + tile * a.y2_ts + lane;
    unsigned err = 0;

    // State(prev) = R^-1 b  (srif.go:223-234)
    T xprev[NS];
    {
        T Rw[NS * NS];
#pragma unroll
        for (int i = 0; i < NS; i++) xprev[i] = sl(st, i);
#pragma unroll
        for (int e = 0; e < NS * NS; e++) Rw[e] = sl(st, NS + e);
        if (lu_solve_inplace<T, NS, 1>(Rw, xprev)) err |= KB_ST_SINGULAR;
    }
    // xBar = Phi xprev; Phi^-1  (srif.go:111-118)
    T xBar[NS], invPhi[NS * NS];
    {
        T Phi[NS * NS];
#pragma unroll
        for (int e = 0; e < NS * NS; e++) Phi[e] = EXT ? __builtin_nontemporal_load(ephi + (int64_t)e * a.ext_ld) : snt(mo, a.L.mo_F + e);
        smv<T, NS, NS>(Phi, xprev, xBar);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) invPhi[i * NS + j] = (i == j) ? T(1) : T(0);
        if (lu_solve_inplace<T, NS, NS>(Phi, invPhi)) err |= KB_ST_SINGULAR;
    }
    // panel A = [[RBar, bBar],[L Htilde, L y]], RBar = R Phi^-1, bBar = RBar xBar (srif.go:115-119, :298-320)
    T A[(NS + NM) * COLS];
    T *es = FULL ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane : nullptr;
#pragma unroll
    for (int i = 0; i < NS; i++) {
        T Ri[NS];
#pragma unroll
        for (int l = 0; l < NS; l++) Ri[l] = sl(st, NS + i * NS + l);  // second read of row i: L2 / Infinity Cache hit
        T bb = T(0);
#pragma unroll
        for (int j = 0; j < NS; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += Ri[l] * invPhi[l * NS + j];
            A[i * COLS + j] = s;
            bb += s * xBar[j];
            if constexpr (FULL) { if (active) ss(es, a.L.es_ppred + i * NS + j, s); }
        }
        A[i * COLS + NS] = bb;
    }
    {
        T Lw[tri(NM)], yv[NM];
#pragma unroll
        for (int e = 0; e < tri(NM); e++) Lw[e] = snt(mo, a.L.mo_LR + e);  // QUIRK srif.go:48: chol_L(R), not its inverse
#pragma unroll
        for (int r = 0; r < NM; r++) {
            const T re = active ? __builtin_nontemporal_load(yr + (int64_t)r * a.y_es) : T(0);
            const T co = active ? __builtin_nontemporal_load(yc + (int64_t)r * a.y2_es) : T(0);
            yv[r] = re - co;
            if constexpr (FULL) { if (active) ss(es, a.L.es_yhat + r, re); }
        }
#pragma unroll
        for (int r = 0; r < NM; r++)
#pragma unroll
            for (int j = 0; j < NS; j++) A[(NS + r) * COLS + j] = T(0);
#pragma unroll
        for (int l = 0; l < NM; l++) {
            T Hl[NS];
#pragma unroll
            for (int j = 0; j < NS; j++) Hl[j] = EXT ? __builtin_nontemporal_load(eh + (int64_t)(l * NS + j) * a.ext_ld) : snt(mo, a.L.mo_H + l * NS + j);
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
            if constexpr (FULL) { if (active) ss(es, a.L.es_dobs + r, s); }
        }
    }
    shouseholder<T, NS, NM>(A);
    T chk = T(0);
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = i; j < COLS; j++) chk += A[i * COLS + j] * T(0);
    if (chk != chk) err |= KB_ST_NONFINITE;
    if (active && !err) {
#pragma unroll
        for (int i = 0; i < NS; i++) ss(st, i, A[i * COLS + NS]);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) ss(st, NS + i * NS + j, j >= i ? A[i * COLS + j] : T(0));
        if constexpr (FULL) {
#pragma unroll
            for (int r = 0; r < NM; r++) ss(es, a.L.es_innov + r, A[(NS + r) * COLS + NS]);
        }
    }
    if (active && err) atomicOr(a.status + fi, err);
}

static bool srif_shape_ok(const StepArgs &a, int NS, int NM) { return a.n == NS && a.p == NM && !a.predict; }

template <typename T, int NS, int NM>
static bool srif_try(const Batch &b, const StepArgs &a) {
    if (!srif_shape_ok(a, NS, NM)) return false;
    const dim3 grid = tile_grid(a.ntiles), block(256);
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
#define KB_S(F_) do { if (a.ext_phi) hipLaunchKernelGGL((srif_reg_kernel<T, NS, NM, F_, true>), grid, block, 0, b.stream, a); \
                      else hipLaunchKernelGGL((srif_reg_kernel<T, NS, NM, F_, false>), grid, block, 0, b.stream, a); } while (0)
    if (full) KB_S(true); else KB_S(false);
#undef KB_S
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
