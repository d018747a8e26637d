// kb_strict.h -- shared by the KB_FLAG_STRICT_SYMCHECK register kernels (kb_vanilla_strict.hip, kb_hybrid_strict.hip).  Those
// translation units are compiled under `#pragma clang fp contract(off)` (set before any include), so what is defined here rounds
// like gonum, the oracle and the statement-order kernels of kb_kinds.hip: products and sums separately.
#pragma once
#include "kb_internal.h"

namespace kb {

// mat64.Dense.Inverse as inverse_lu_rt (kb_device.h) does it, on a register matrix: partial pivoting with ONE exchange per column
// (the pivot row found first, then swapped in by selects -- no dynamic register indexing), the same elimination and substitution
// order, the same condition test.  Rows / columns >= nreal are identity padding and stay out of the norms.
template <typename T, int P>
__device__ __forceinline__ bool inverse_strict(const T (&Ain)[P * P], T (&X)[P * P], int nreal) {
    T a[P * P], b[P * P];
    T anorm = T(0);
#pragma unroll
    for (int i = 0; i < P; i++) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < P; j++) {
            a[i * P + j] = Ain[i * P + j];
            b[i * P + j] = (i == j) ? T(1) : T(0);
            if (j < nreal) s += fabs(Ain[i * P + j]);
        }
        if (i < nreal) anorm = (s > anorm || s != s) ? s : anorm;
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < P; j++) {
        int jp = j;
        T best = fabs(a[j * P + j]);
#pragma unroll
        for (int r = j + 1; r < P; r++) {
            const bool gt = (r < nreal) && fabs(a[r * P + j]) > best;
            best = gt ? fabs(a[r * P + j]) : best;
            jp = gt ? r : jp;
        }
#pragma unroll
        for (int r = j + 1; r < P; r++) {
            const bool sw = jp == r;
#pragma unroll
            for (int c = 0; c < P; c++) {
                const T t0 = a[j * P + c], t1 = a[r * P + c];
                a[j * P + c] = sw ? t1 : t0;
                a[r * P + c] = sw ? t0 : t1;
                const T u0 = b[j * P + c], u1 = b[r * P + c];
                b[j * P + c] = sw ? u1 : u0;
                b[r * P + c] = sw ? u0 : u1;
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
            for (int c = 0; c < P; c++) b[r * P + c] -= l * b[j * P + c];
        }
    }
    T inorm = T(0);
#pragma unroll
    for (int i = P - 1; i >= 0; i--) {
        const T rd = T(1) / a[i * P + i];
#pragma unroll
        for (int c = 0; c < P; c++) {
            T s = b[i * P + c];
#pragma unroll
            for (int k = i + 1; k < P; k++) s -= a[i * P + k] * X[k * P + c];
            X[i * P + c] = s * rd;
        }
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
        T s = T(0);
#pragma unroll
        for (int c = 0; c < P; c++)
            if (c < nreal) s += fabs(X[i * P + c]);
        if (i < nreal) inorm = (s > inorm || s != s) ? s : inorm;
    }
    const T cond = anorm * inorm;
    return bad || !(cond <= T(1e16));
}

}  // namespace kb
