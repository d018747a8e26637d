// kb_static.h -- compile-time-dimension small dense algebra on register-resident arrays
// (tightly packed row-major, every loop fully unrolled, no runtime indexing => no scratch).
// Same algorithms and operation order as kb_dense.h / the oracle.
#pragma once
#include "kb_device.h"

namespace kb {

// C(R x C) = A(R x K) B(K x C)
template <typename T, int R, int K, int C>
__device__ __forceinline__ void smm_nn(const T (&A)[R * K], const T (&B)[K * C], T (&Cm)[R * C]) {
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
        for (int j = 0; j < C; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < K; l++) s += A[i * K + l] * B[l * C + j];
            Cm[i * C + j] = s;
        }
}
// C(R x C) = A(R x K) B(C x K)^T
template <typename T, int R, int K, int C>
__device__ __forceinline__ void smm_nt(const T (&A)[R * K], const T (&B)[C * K], T (&Cm)[R * C]) {
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
        for (int j = 0; j < C; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < K; l++) s += A[i * K + l] * B[j * K + l];
            Cm[i * C + j] = s;
        }
}
template <typename T, int R, int C>
__device__ __forceinline__ void smv(const T (&A)[R * C], const T (&x)[C], T (&y)[R]) {
#pragma unroll
    for (int i = 0; i < R; i++) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < C; j++) s += A[i * C + j] * x[j];
        y[i] = s;
    }
}

// Dgeqr2 on a register panel a[M*N] (row-major), in place.  ACTIVE(k, r) tells at compile time
// whether row r can hold a non-zero in column k when column k is eliminated (rows that are
// structurally zero there are skipped: they would contribute exact zeros to the norm and to
// every update, so results are unchanged).  On return the upper triangle holds R.
struct AllRowsActive {
    static constexpr bool active(int, int) { return true; }
};
// FASTDIV: f from recip() (kb_device.h: within an ulp of the quotient, a third of its instructions) -- the time-fused SquareRoot kernel,
// which is bound by instruction issue and does not promise the one-step kernel's bits anyway.
template <typename T, int M, int N, typename ACT = AllRowsActive, bool FASTDIV = false>
__device__ __forceinline__ void sqr_r(T (&a)[M * N]) {
    // Every multiply-add below is an EXPLICIT fma: under -ffp-contract=fast the compiler is free to contract `u0 a + x y` either way round,
    // and it chose differently in two instantiations of the same kernel (the one-step and the time-fused SquareRoot kernel ended a last place
    // apart from the second step on; found in round 6 by bisecting `#pragma clang fp contract(off)` over the blocks of the step: with it
    // in THIS function alone the two agree bit for bit -- scripts/diag_sqrt_fused.py, profiles/NOTES.md).  Spelled out, there is no choice left.
    constexpr int KMAX = M < N ? M : N;
#pragma unroll
    for (int i = 0; i < KMAX; i++) {
        T xnorm2 = T(0);
#pragma unroll
        for (int r = i + 1; r < M; r++)
            if (ACT::active(i, r)) xnorm2 = __builtin_fma(a[r * N + i], a[r * N + i], xnorm2);
        // Dlarfg: beta = -sign(alpha) * dlapy2(alpha, xnorm).  dlapy2 only guards against overflow of the squares;
        // the panels here are covariance square roots, so the plain sqrt of the sum is used (one rounding apart).
        // The reflector is applied in its unnormalised form H = I + f u u^T, u = (alpha - beta, x),
        // f = 1 / (beta (alpha - beta)): the same matrix as LAPACK's I - tau v v^T (v = u / u0,
        // tau = (beta - alpha) / beta) with one division per column instead of two and no scaling pass.
        const bool refl = (M - i > 1) && (xnorm2 != T(0));
        const T alpha = a[i * N + i];
        const T beta = -copysign(sqrt(__builtin_fma(alpha, alpha, xnorm2)), alpha);
        const T u0 = alpha - beta;
        const T f = refl ? (FASTDIV ? recip(beta * u0) : T(1) / (beta * u0)) : T(0);
        a[i * N + i] = refl ? beta : alpha;
#pragma unroll
        for (int c = i + 1; c < N; c++) {
            T s = u0 * a[i * N + c];
#pragma unroll
            for (int r = i + 1; r < M; r++)
                if (ACT::active(i, r)) s = __builtin_fma(a[r * N + i], a[r * N + c], s);
            const T fs = f * s;
            a[i * N + c] = __builtin_fma(fs, u0, a[i * N + c]);
#pragma unroll
            for (int r = i + 1; r < M; r++)
                if (ACT::active(i, r)) a[r * N + c] = __builtin_fma(fs, a[r * N + i], a[r * N + c]);
        }
    }
}

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
            if (__any(sw)) {   // wave-uniform: the exchange code (2 selects per element) only runs when some lane needs it
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

// helper.go:142-172 HouseholderTransf(A, n, m) on a register panel A[(NN+MM)*(NN+1)]
template <typename T, int NN, int MM>
__device__ __forceinline__ void shouseholder(T (&A)[(NN + MM) * (NN + 1)]) {
    constexpr int ROWS = NN + MM, COLS = NN + 1;
#pragma unroll
    for (int k = 0; k < NN; k++) {
        T sigma = T(0);
#pragma unroll
        for (int i = k; i < ROWS; i++) sigma += A[i * COLS + k] * A[i * COLS + k];
        const T akk = A[k * COLS + k];
        const T sgn = (akk == T(0) || fabs(akk) <= T(1e-12)) ? T(1) : copysign(T(1), akk);  // helper.go:133-138 Sign: v / |v|
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
#pragma unroll
        for (int i = k + 1; i < ROWS; i++) A[i * COLS + k] = T(0);
    }
}

}  // namespace kb
