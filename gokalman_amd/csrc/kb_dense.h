// kb_dense.h -- run-time-dimension small dense algebra on per-lane private arrays
// (row-major with a compile-time leading dimension).  Used by the generic kernels, the
// constructor ("derive") kernels and the lazy Estimate getters.  Algorithms restate the
// gonum/LAPACK routines the reference calls, in the same order as oracle/gokalman_oracle.c.
#pragma once
#include "kb_device.h"

namespace kb {

// C(r x c) = A(r x k) B(k x c)
template <typename T, int LA, int LB, int LC>
__device__ inline void mm_nn(int r, int k, int c, const T *A, const T *B, T *C) {
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) {
            T s = T(0);
            for (int l = 0; l < k; l++) s += A[i * LA + l] * B[l * LB + j];
            C[i * LC + j] = s;
        }
}
// C(r x c) = A(r x k) B(c x k)^T
template <typename T, int LA, int LB, int LC>
__device__ inline void mm_nt(int r, int k, int c, const T *A, const T *B, T *C) {
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) {
            T s = T(0);
            for (int l = 0; l < k; l++) s += A[i * LA + l] * B[j * LB + l];
            C[i * LC + j] = s;
        }
}
// C(r x c) = A(k x r)^T B(k x c)
template <typename T, int LA, int LB, int LC>
__device__ inline void mm_tn(int r, int k, int c, const T *A, const T *B, T *C) {
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) {
            T s = T(0);
            for (int l = 0; l < k; l++) s += A[l * LA + i] * B[l * LB + j];
            C[i * LC + j] = s;
        }
}
// C(r x c) = A(k x r)^T B(c x k)^T
template <typename T, int LA, int LB, int LC>
__device__ inline void mm_tt(int r, int k, int c, const T *A, const T *B, T *C) {
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) {
            T s = T(0);
            for (int l = 0; l < k; l++) s += A[l * LA + i] * B[j * LB + l];
            C[i * LC + j] = s;
        }
}
template <typename T, int LA>
__device__ inline void mv_n(int r, int c, const T *A, const T *x, T *y) {
    for (int i = 0; i < r; i++) {
        T s = T(0);
        for (int j = 0; j < c; j++) s += A[i * LA + j] * x[j];
        y[i] = s;
    }
}
template <typename T, int LA>
__device__ inline void mv_t(int r, int c, const T *A, const T *x, T *y) {  // y = A^T x
    for (int j = 0; j < c; j++) {
        T s = T(0);
        for (int i = 0; i < r; i++) s += A[i * LA + j] * x[i];
        y[j] = s;
    }
}

// mat64.Cholesky (Dpotrf, reads the upper triangle) -> L lower (L = U^T).  false = not PD.
template <typename T, int LD>
__device__ inline bool cholesky_lower_rt(int n, const T *A, T *L) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) L[i * LD + j] = T(0);
    for (int j = 0; j < n; j++) {
        T ajj = A[j * LD + j];
        for (int k = 0; k < j; k++) ajj -= L[j * LD + k] * L[j * LD + k];
        if (!(ajj > T(0))) return false;
        ajj = sqrt(ajj);
        L[j * LD + j] = ajj;
        for (int i = j + 1; i < n; i++) {
            T s = A[j * LD + i];
            for (int k = 0; k < j; k++) s -= L[j * LD + k] * L[i * LD + k];
            L[i * LD + j] = s / ajj;
        }
    }
    return true;
}

// mat64.QR.Factorize + RFromQR: Dgeqr2 (Dlarfg / Dlarf), in place on a (m x n), LD leading dim.
// On return the upper triangle holds R; entries below the diagonal are garbage (reflectors).
template <typename T, int LD>
__device__ inline void qr_r_rt(int m, int n, T *a) {
    T w[LD];
    const int kmax = m < n ? m : n;
    for (int i = 0; i < kmax; i++) {
        T tau = T(0);
        if (m - i > 1) {
            T xnorm = T(0);
            for (int r = i + 1; r < m; r++) xnorm += a[r * LD + i] * a[r * LD + i];
            xnorm = sqrt(xnorm);
            if (xnorm != T(0)) {
                const T alpha = a[i * LD + i];
                const T beta = -copysign(hypot(alpha, xnorm), alpha);
                tau = (beta - alpha) / beta;
                const T sc = T(1) / (alpha - beta);
                for (int r = i + 1; r < m; r++) a[r * LD + i] *= sc;
                a[i * LD + i] = beta;
            }
        }
        if (i < n - 1 && tau != T(0)) {
            for (int c = i + 1; c < n; c++) {
                T s = a[i * LD + c];
                for (int r = i + 1; r < m; r++) s += a[r * LD + i] * a[r * LD + c];
                w[c] = s;
            }
            for (int c = i + 1; c < n; c++) {
                a[i * LD + c] -= tau * w[c];
                for (int r = i + 1; r < m; r++) a[r * LD + c] -= tau * a[r * LD + i] * w[c];
            }
        }
    }
}

// helper.go:133-138 Sign
template <typename T>
__device__ __forceinline__ T sign_ref(T v) {
    return (v == T(0) || fabs(v) <= T(1e-12)) ? T(1) : v / fabs(v);
}

// helper.go:142-172 HouseholderTransf(A, n, m): A is (m+n) x (n+1) with leading dim LD, in place.
template <typename T, int LD, int ROWS>
__device__ inline void householder_transf_rt(T *A, int n, int m) {
    const int rows = m + n;
    T u[ROWS];
    for (int k = 0; k < n; k++) {
        T sigma = T(0);
        for (int i = k; i < rows; i++) sigma += A[i * LD + k] * A[i * LD + k];
        sigma = sqrt(sigma) * sign_ref(A[k * LD + k]);
        u[k] = A[k * LD + k] + sigma;
        A[k * LD + k] = -sigma;
        for (int i = k + 1; i < rows; i++) u[i] = A[i * LD + k];
        const T beta = T(1) / (sigma * u[k]);
        for (int j = k + 1; j < n + 1; j++) {
            T gamma = T(0);
            for (int i = k; i < rows; i++) gamma += u[i] * A[i * LD + j];
            gamma *= beta;
            for (int i = k; i < rows; i++) A[i * LD + j] = A[i * LD + j] - gamma * u[i];
        }
        for (int i = k + 1; i < rows; i++) A[i * LD + k] = T(0);
    }
}

}  // namespace kb
