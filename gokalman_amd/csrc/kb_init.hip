// kb_init.hip -- constructor / setter side-effect arithmetic ("derive" ops) and the lazy
// Estimate getters, all on the device (one filter per lane, run-time dimensions).
//
//   NewSquareRoot     squareroot.go:33-49   S0 = chol_L(P0);  SetNoise :100-114  chol_L(Q), chol_L(R)
//   NewInformation    information.go:39-50  F^-1, Q^-1, R^-1;  FromState :65-81  I0 = P0^-1, i0 = I0 x0
//   NewSRIF           srif.go:20-45         I0 = diag(1/P0_ii), R0 = chol_L(I0), b0 = R0 x0, L = chol_L(R)
//   AWGN              noise.go:145-159      L_Q, L_R (distmv.NewNormal needs PD matrices)
//   getters           squareroot.go:317-340, information.go:257-316, srif.go:223-281
#include "kb_dense.h"
#include "kb_internal.h"

namespace kb {

enum DeriveOp {
    OP_SQRT_P0 = 1,    // state mat (packed sym P0) -> packed lower chol, in place
    OP_CHOL_Q = 2,     // model Q -> mo_LQ
    OP_CHOL_R = 3,     // model R (dim rp) -> mo_LR
    OP_INV_F = 4,      // model F -> mo_Finv
    OP_INV_Q = 5,      // model Q -> mo_Qinv (packed: the upper triangle of the computed inverse)
    OP_INV_R = 6,      // model R (dim rp) -> mo_Rinv (packed)
    OP_INFO_FROM_STATE = 7,
    OP_SRIF_INIT = 8
};


template <typename T, int LD>
__global__ void __launch_bounds__(64) derive_kernel(void *state_, void *model_, int64_t N, Layout L, int op, int rp,
                                                    int *fail_count) {
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t fi = tile * KB_TILE + lane;
    if (fi >= N) return;
    T *st = (T *)state_ + tile * ((int64_t)KB_TILE * L.st_elems) + lane;
    T *mo = (T *)model_ + tile * ((int64_t)KB_TILE * L.mo_elems) + lane;
    const int n = L.n;
    T A[LD * LD], B[LD * LD];
    bool fail = false;
    switch (op) {
    case OP_SQRT_P0: {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) A[i * LD + j] = ldt(st, L.st_mat + symi(i, j));
        fail = !cholesky_lower_rt<T, LD>(n, A, B);
        for (int i = 0; i < n; i++)
            for (int k = 0; k <= i; k++) stt(st, L.st_mat + symi(k, i), fail ? T(0) : B[i * LD + k]);
        break;
    }
    case OP_CHOL_Q: {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) A[i * LD + j] = ldt(mo, L.mo_Q + symi(i, j));
        fail = !cholesky_lower_rt<T, LD>(n, A, B);
        for (int i = 0; i < n; i++)
            for (int k = 0; k <= i; k++) stt(mo, L.mo_LQ + symi(k, i), fail ? T(0) : B[i * LD + k]);
        break;
    }
    case OP_CHOL_R: {
        for (int i = 0; i < rp; i++)
            for (int j = 0; j < rp; j++) A[i * LD + j] = ldt(mo, L.mo_R + symi(i, j));
        fail = !cholesky_lower_rt<T, LD>(rp, A, B);
        for (int i = 0; i < rp; i++)
            for (int k = 0; k <= i; k++) stt(mo, L.mo_LR + symi(k, i), fail ? T(0) : B[i * LD + k]);
        break;
    }
    case OP_INV_F: {  // errors are only printed by the reference (information.go:39-41)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) A[i * LD + j] = ldt(mo, L.mo_F + i * n + j);
        inverse_lu_rt<T, LD>(n, A, B);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) stt(mo, L.mo_Finv + i * n + j, B[i * LD + j]);
        break;
    }
    case OP_INV_Q: {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) A[i * LD + j] = ldt(mo, L.mo_Q + symi(i, j));
        inverse_lu_rt<T, LD>(n, A, B);
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) stt(mo, L.mo_Qinv + symi(i, j), B[i * LD + j]);
        break;
    }
    case OP_INV_R: {
        for (int i = 0; i < rp; i++)
            for (int j = 0; j < rp; j++) A[i * LD + j] = ldt(mo, L.mo_R + symi(i, j));
        inverse_lu_rt<T, LD>(rp, A, B);
        for (int i = 0; i < rp; i++)
            for (int j = i; j < rp; j++) stt(mo, L.mo_Rinv + symi(i, j), B[i * LD + j]);
        break;
    }
    case OP_INFO_FROM_STATE: {  // information.go:65-81
        T x[LD];
        for (int i = 0; i < n; i++) x[i] = ldt(st, L.st_vec + i);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) A[i * LD + j] = ldt(st, L.st_mat + symi(i, j));
        const bool bad = inverse_lu_rt<T, LD>(n, A, B);
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) { const T v = bad ? T(0) : B[i * LD + j]; B[i * LD + j] = v; B[j * LD + i] = v; }
        for (int i = 0; i < n; i++) {
            T s = T(0);
            for (int j = 0; j < n; j++) s += B[i * LD + j] * x[j];
            stt(st, L.st_vec + i, s);
        }
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) stt(st, L.st_mat + symi(i, j), B[i * LD + j]);
        break;
    }
    case OP_SRIF_INIT: {  // srif.go:20-35: state block holds x0 | P0 (full) on entry, b0 | R0 on exit
        T x[LD];
        for (int i = 0; i < n; i++) x[i] = ldt(st, L.st_vec + i);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) A[i * LD + j] = (i == j) ? T(1) / ldt(st, L.st_mat + i * n + i) : T(0);
        fail = !cholesky_lower_rt<T, LD>(n, A, B);
        for (int i = 0; i < n; i++) {
            T s = T(0);
            for (int j = 0; j < n; j++) s += B[i * LD + j] * x[j];
            stt(st, L.st_vec + i, fail ? T(0) : s);
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) stt(st, L.st_mat + i * n + j, fail ? T(0) : B[i * LD + j]);
        break;
    }
    }
    if (fail) atomicAdd(fail_count, 1);
}

static int run_derive(Batch &b, int op, int rp, int *fails) {
    int *d_cnt = nullptr;
    KB_HIP(hipMalloc((void **)&d_cnt, sizeof(int)));
    KB_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int), b.stream));
    const dim3 grid((unsigned)b.ntiles), block(64);
    const int d = b.n > b.pmax ? b.n : b.pmax;
#define KB_D(TT, LDD) hipLaunchKernelGGL((derive_kernel<TT, LDD>), grid, block, 0, hs.stream, b.d_state, b.d_model, b.N, b.L, op, rp, d_cnt)
    {
        const HeavyScope hs(b, d > 8);   // LD = 16: scratch-heavy, see kb_internal.h
        if (b.dtype == KB_F64) { if (d <= 4) KB_D(double, 4); else if (d <= 8) KB_D(double, 8); else KB_D(double, 16); }
        else                   { if (d <= 4) KB_D(float, 4);  else if (d <= 8) KB_D(float, 8);  else KB_D(float, 16); }
    }
#undef KB_D
    int cnt = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost, b.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b.stream);
    (void)hipFree(d_cnt);
    if (e != hipSuccess) return hip_fail(e, "derive_kernel");
    *fails += cnt;
    return KB_OK;
}

int launch_init(Batch &b, int *not_pd) {
    *not_pd = 0;
    int rc = KB_OK;
    const int rp = b.r_p;
    switch (b.kind) {
    case KB_VANILLA:
    case KB_VANILLA_PREDICT:
    case KB_HYBRID:
        break;
    case KB_SQUAREROOT:
        if ((rc = run_derive(b, OP_SQRT_P0, rp, not_pd))) return rc;
        if ((rc = run_derive(b, OP_CHOL_Q, rp, not_pd))) return rc;
        if ((rc = run_derive(b, OP_CHOL_R, rp, not_pd))) return rc;
        b.sqrt_p = rp;
        break;
    case KB_INFORMATION: {
        int ignore = 0;
        if (b.flags & KB_FLAG_INFO_FROM_STATE)
            if ((rc = run_derive(b, OP_INFO_FROM_STATE, rp, &ignore))) return rc;
        if ((rc = run_derive(b, OP_INV_F, rp, &ignore))) return rc;
        if ((rc = run_derive(b, OP_INV_Q, rp, &ignore))) return rc;
        if ((rc = run_derive(b, OP_INV_R, rp, &ignore))) return rc;
        b.rinv_p = rp;
        break;
    }
    case KB_SRIF:
        if ((rc = run_derive(b, OP_SRIF_INIT, rp, not_pd))) return rc;
        if ((rc = run_derive(b, OP_CHOL_R, rp, not_pd))) return rc;
        b.sqrt_p = rp;
        break;
    }
    if (b.noise_kind == KB_NOISE_AWGN && b.kind != KB_SQUAREROOT) {
        if (b.kind != KB_HYBRID && b.kind != KB_SRIF)
            if ((rc = run_derive(b, OP_CHOL_Q, rp, not_pd))) return rc;
        if (b.kind != KB_SRIF)
            if ((rc = run_derive(b, OP_CHOL_R, rp, not_pd))) return rc;
    }
    return KB_OK;
}

int launch_refresh(Batch &b, int field, int *not_pd) {
    *not_pd = 0;
    int rc = KB_OK, ignore = 0;
    const int rp = b.r_p;
    if (field == KB_F && b.kind == KB_INFORMATION) return run_derive(b, OP_INV_F, rp, &ignore);
    const bool chol = (b.kind == KB_SQUAREROOT) || (b.noise_kind == KB_NOISE_AWGN && (b.kind == KB_VANILLA || b.kind == KB_VANILLA_PREDICT || b.kind == KB_INFORMATION));
    if (chol && field == KB_Q) rc = run_derive(b, OP_CHOL_Q, rp, not_pd);
    if (chol && field == KB_R) {
        rc = run_derive(b, OP_CHOL_R, rp, not_pd);
        if (b.kind == KB_SQUAREROOT) b.sqrt_p = rp;
    }
    return rc;
}

// -------------------------------------------------------------------------------------
// lazy getters: out block = x [n] | P packed [tri(n)]
// -------------------------------------------------------------------------------------
template <typename T, int LD>
__global__ void __launch_bounds__(64) materialise_kernel(const void *src_, int src_elems, int vec_off, int mat_off,
                                                         const void *vec_src_, int vec_src_elems, int vec_src_off,
                                                         void *out_, uint32_t *status, int64_t N, int n, int kind, int pred) {
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t fi = tile * KB_TILE + lane;
    if (fi >= N) return;
    const T *src = (const T *)src_ + tile * ((int64_t)KB_TILE * src_elems) + lane;
    const T *vsrc = (const T *)vec_src_ + tile * ((int64_t)KB_TILE * vec_src_elems) + lane;
    T *out = (T *)out_ + tile * ((int64_t)KB_TILE * (n + tri(n))) + lane;
    T M[LD * LD], W[LD * LD], v[LD], x[LD];
    (void)vec_off;
    for (int i = 0; i < n; i++) v[i] = ldt(vsrc, vec_src_off + i);
    if (kind == KB_SQUAREROOT) {
        // S lower packed (posterior) or S- = Uc upper packed (predicted, squareroot.go:185 quirk)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                const bool nz = pred ? (j >= i) : (j <= i);
                M[i * LD + j] = nz ? ldt(src, mat_off + symi(i, j)) : T(0);
            }
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) {
                T s = T(0);
                for (int k = 0; k < n; k++) s += M[i * LD + k] * M[j * LD + k];
                stt(out, n + symi(i, j), s);
            }
        for (int i = 0; i < n; i++) stt(out, i, v[i]);
        return;
    }
    if (kind == KB_INFORMATION) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) M[i * LD + j] = ldt(src, mat_off + symi(i, j));
        const bool bad = inverse_lu_rt<T, LD>(n, M, W);  // information.go:284-288: zeros + warning
        if (bad && !pred) atomicOr(status + fi, KB_ST_INFO_NOT_INVERTIBLE);
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) { const T val = bad ? T(0) : W[i * LD + j]; W[i * LD + j] = val; W[j * LD + i] = val; }
        for (int i = 0; i < n; i++) {
            T s = T(0);
            for (int j = 0; j < n; j++) s += W[i * LD + j] * v[j];
            x[i] = s;
        }
    } else if (kind == KB_BATCH_LS) {  // batch.go:64-79 Solve: P0 = AsSymDense(inverse(Lambda)), xHat0 = P0 N
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) M[i * LD + j] = ldt(src, mat_off + i * n + j);
        bool bad = inverse_lu_rt<T, LD>(n, M, W);
        bool sym = true;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                if (i != j) sym = sym && sym_close(W[j * LD + i], W[i * LD + j]);
        if (bad) atomicOr(status + fi, (unsigned)KB_ST_SINGULAR);
        else if (!sym) atomicOr(status + fi, (unsigned)KB_ST_ASYMMETRIC);
        bad = bad || !sym;
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) { const T val = bad ? T(0) : W[i * LD + j]; W[i * LD + j] = val; W[j * LD + i] = val; }
        for (int i = 0; i < n; i++) {
            T s = T(0);
            for (int j = 0; j < n; j++) s += W[i * LD + j] * v[j];
            x[i] = s;
        }
    } else {  // KB_SRIF: x = R^-1 b, P = R^-1 R^-T (srif.go:223-281)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) M[i * LD + j] = ldt(src, mat_off + i * n + j);
        const bool bad = inverse_lu_rt<T, LD>(n, M, W);
        if (bad && !pred) atomicOr(status + fi, KB_ST_INFO_NOT_INVERTIBLE);
        for (int i = 0; i < n; i++) {
            T s = T(0);
            for (int j = 0; j < n; j++) s += W[i * LD + j] * v[j];
            x[i] = bad ? T(0) : s;
        }
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) {
                T s = T(0);
                for (int k = 0; k < n; k++) s += W[i * LD + k] * W[j * LD + k];
                M[i * LD + j] = bad ? T(0) : s;
            }
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) W[i * LD + j] = M[i * LD + j];
    }
    for (int i = 0; i < n; i++) stt(out, i, x[i]);
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) stt(out, n + symi(i, j), W[i * LD + j]);
}

int launch_materialise(const Batch &b, const void *src_block, bool pred, void *out_block) {
    const int n = b.n;
    const int src_elems = pred ? b.L.es_elems : b.L.st_elems;
    const int mat_off = pred ? b.L.es_ppred : b.L.st_mat;
    const dim3 grid((unsigned)b.ntiles), block(64);
#define KB_M(TT, LDD) hipLaunchKernelGGL((materialise_kernel<TT, LDD>), grid, block, 0, hs.stream, src_block, src_elems, \
                                         b.L.st_vec, mat_off, (const void *)b.d_state, b.L.st_elems, b.L.st_vec, out_block, b.d_status, b.N, n, b.kind, pred ? 1 : 0)
    {
        const HeavyScope hs(b, n > 8);   // LD = 16: scratch-heavy, see kb_internal.h
        if (b.dtype == KB_F64) { if (n <= 4) KB_M(double, 4); else if (n <= 8) KB_M(double, 8); else KB_M(double, 16); }
        else                   { if (n <= 4) KB_M(float, 4);  else if (n <= 8) KB_M(float, 8);  else KB_M(float, 16); }
    }
#undef KB_M
    KB_HIP(hipGetLastError());
    return KB_OK;
}

// SmoothAll (hybrid.go:209-238, srif.go:165-192): backward sweep x_k = S x_{k+1}, P_k = sym(S P_{k+1} S^T),
// S = inverse(Phi_{k+1}).  xp_block holds the last estimate's State() | Covariance() (packed); phis is the
// caller's planar history [steps][n*n][ld]; outputs are planar [steps][n][ld] and [steps][n*n][ld].
template <typename T, int LD>
__global__ void __launch_bounds__(64) smooth_kernel(const T *xp_block, int xp_elems, int vec_off, int mat_off, const T *phis, int64_t ld,
                                                    int steps, T *x_out, T *P_out, uint32_t *status, int64_t N, int n) {
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t fi = tile * KB_TILE + lane;
    if (fi >= N) return;
    const T *src = xp_block + tile * ((int64_t)KB_TILE * xp_elems) + lane;
    T x[LD], P[LD * LD], S[LD * LD], W[LD * LD], A[LD * LD];
    for (int i = 0; i < n; i++) x[i] = ldt(src, vec_off + i);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) P[i * LD + j] = ldt(src, mat_off + symi(i, j));
    unsigned err = 0;
    for (int k = steps - 1; k >= 0; k--) {
        for (int i = 0; i < n; i++) x_out[((int64_t)k * n + i) * ld + fi] = x[i];
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) P_out[((int64_t)k * n * n + i * n + j) * ld + fi] = P[i * LD + j];
        if (k == 0 || err) { if (err && k > 0) continue; break; }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) A[i * LD + j] = phis[((int64_t)k * n * n + i * n + j) * ld + fi];   // Phi of estimate k (= k+1 of the pair)
        if (inverse_lu_rt<T, LD>(n, A, S)) { err |= KB_ST_SINGULAR; continue; }                          // "provided STM is not invertible"
        mm_nn<T, LD, LD, LD>(n, n, n, S, P, W);
        mm_nt<T, LD, LD, LD>(n, n, n, W, S, A);
        T xn[LD];
        mv_n<T, LD>(n, n, S, x, xn);
        bool sym = true;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                if (i != j) sym = sym && sym_close(A[j * LD + i], A[i * LD + j]);
        if (!sym) { err |= KB_ST_ASYMMETRIC; continue; }
        for (int i = 0; i < n; i++) x[i] = xn[i];
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) { P[i * LD + j] = A[i * LD + j]; P[j * LD + i] = A[i * LD + j]; }
    }
    if (err) atomicOr(status + fi, err);
}

int launch_smooth(const Batch &b, const void *xp_block, int xp_elems, int vec_off, int mat_off, const void *phis, int64_t ld, int steps,
                  void *x_out, void *P_out) {
    const dim3 grid((unsigned)b.ntiles), block(64);
    const int n = b.n;
#define KB_SM(TT, LDD) hipLaunchKernelGGL((smooth_kernel<TT, LDD>), grid, block, 0, hs.stream, (const TT *)xp_block, xp_elems, vec_off, mat_off, \
                                          (const TT *)phis, ld, steps, (TT *)x_out, (TT *)P_out, b.d_status, b.N, n)
    {
        const HeavyScope hs(b, n > 8);   // LD = 16: scratch-heavy, see kb_internal.h
        if (b.dtype == KB_F64) { if (n <= 4) KB_SM(double, 4); else if (n <= 8) KB_SM(double, 8); else KB_SM(double, 16); }
        else                   { if (n <= 4) KB_SM(float, 4);  else if (n <= 8) KB_SM(float, 8);  else KB_SM(float, 16); }
    }
#undef KB_SM
    KB_HIP(hipGetLastError());
    return KB_OK;
}

// Estimate.IsWithinNsigma (vanilla.go:231-239): |x_i| <= N sqrt(P_ii) for all i
template <typename T>
__global__ void within_kernel(const T *xp, int elems, int vec_off, int mat_off, int n, double nsigma, int64_t N, uint8_t *out) {
    const int64_t fi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (fi >= N) return;
    const T *s = xp + (fi / KB_TILE) * ((int64_t)KB_TILE * elems) + (fi % KB_TILE);
    bool ok = true;
    for (int i = 0; i < n; i++) {
        const double ns = nsigma * sqrt((double)ldt(s, mat_off + symi(i, i)));
        const double x = (double)ldt(s, vec_off + i);
        if (x > ns || x < -ns) ok = false;
    }
    out[fi] = ok ? 1 : 0;
}

int launch_within_nsigma(const Batch &b, const void *xp_block, double nsigma, uint8_t *d_out) {
    const bool lazy = (b.kind == KB_SQUAREROOT || b.kind == KB_INFORMATION || b.kind == KB_SRIF);
    const int elems = lazy ? b.n + tri(b.n) : b.L.st_elems;
    const int vec_off = lazy ? 0 : b.L.st_vec, mat_off = lazy ? b.n : b.L.st_mat;
    const unsigned blocks = (unsigned)((b.N + 255) / 256);
    if (b.dtype == KB_F64)
        hipLaunchKernelGGL(within_kernel<double>, dim3(blocks), dim3(256), 0, b.stream, (const double *)xp_block, elems, vec_off, mat_off, b.n, nsigma, b.N, d_out);
    else
        hipLaunchKernelGGL(within_kernel<float>, dim3(blocks), dim3(256), 0, b.stream, (const float *)xp_block, elems, vec_off, mat_off, b.n, nsigma, b.N, d_out);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
