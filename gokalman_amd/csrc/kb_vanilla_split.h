// kb_vanilla_split.h -- Vanilla.Update (vanilla.go:128-220) for 8 < n <= 16 with ONE FILTER SPLIT OVER L LANES.
//
// Why: with one filter per lane a 12-state step needs F (144) + P (78) + P- (78) + A (144) + ... values per lane, far beyond the
// 128 doubles that leave two waves on a SIMD; the run-time-dimension kernel (kb_vanilla.hip) keeps them in 10-31 KB of scratch per
// lane and runs at a few % of the HBM roof.  Here a wave owns 64 / L filters (L = 4 at n = 12: a quarter of an AoSoA-64 tile):
//
//   lane mapping   lane = q * (64 / L) + f: lanes with the same q read 64 / L consecutive filters of one element, i.e. one
//                  contiguous 128-byte segment in fp64 at L = 4 (the mapping kb_srif_pair.h measured at the rate of the
//                  one-filter-per-lane stream; the interleaved mapping lane = L f + q loses a quarter of the bandwidth).
//   rows           lane q owns rows i = q, q + L, q + 2 L, ... of every n-row matrix (cyclic, so that the triangles balance): its
//                  rows of F, of T = F P, of P-, of P- H^T, K, A = I - K H, A P- and P+ live in ITS registers -- n / L rows.
//   broadcast      a product (own rows) x (whole matrix) needs the second operand in every lane of the filter: that operand sits
//                  in LDS, [element][filter] (conflict-free: the L lanes of a filter read the same word), one n x n region per
//                  wave that is reused phase by phase:  P (packed) -> F -> P- (packed) | H -> K.  150 doubles per filter at 12/6:
//                  19.2 KB per wave, eight waves per CU = two per SIMD.
//   sums over rows S = H (P- H^T) + R and H x- run over ALL rows: each lane sums its own rows and the L partial sums are added
//                  with v_permlane32_swap / v_permlane16_swap (gfx950), identically in every lane; (H P- H^T + R)^-1 (p x p) is
//                  then formed redundantly by the L lanes up to p = 6, and ONCE per filter by its lanes at p = 7, 8 (dist_inverse
//                  below: Gauss-Jordan by columns, the pivot column handed over through LDS).
//   LDS slots      the regions read as contiguous runs (rows of F, columns of H, rows of P- H^T and K) hold TWO elements per lane
//                  (16 bytes): ds_read_b128 at twice the array rate of the ds_read2_b64 pairs the compiler forms from 8-byte slots.
//
// Arithmetic: the reference's statements in the reference's order, with these rounding-level differences: sums over all rows are
// added as L partial sums; S is the mirrored upper triangle; P- enters P- H^T with the lane's own computed entries right of
// column L r and with the mirrored upper triangle left of it; and the Joseph update is evaluated as
//     AP = P- - K (P- H^T)^T  [= (I - K H) P-],   P+ = AP - (AP H^T - K R) K^T      [= A P- A^T + K R K^T, vanilla.go:197-205]
// i.e. both multiplications by A = I - K H distributed: no n x n operand has to be formed or broadcast for them (K, H and P- H^T
// are n x p), 650 FMAs per lane less than A = I - K H, A P-, (A P-) A^T.  The rounding error of AP is of the size the reference's
// A P- has (there the cancellation 1 - (K H)_ii sits inside A, here in the subtraction: both leave eps |P-|), and AP H^T is formed
// from the COMPUTED AP, so that error is multiplied by A^T as in the reference's product -- the property of the Joseph form.
// Failure semantics as in kb_vanilla_reg.h: a filter whose S is singular / ill-conditioned or whose result is non-finite keeps
// its previous estimate and gets a status bit (the reference's (nil, err)).
#pragma once
#include <type_traits>

#include "kb_internal.h"
#include "kb_vanilla_reg.h"

namespace kb {

// ---- sums over the L lanes of a filter (lanes 64 / L apart) -------------------------------------------------------------
__device__ __forceinline__ unsigned other32(unsigned x) {   // the value held by the lane 32 away
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // r[0] = [x.lo | x.lo], r[1] = [x.hi | x.hi]
    return (threadIdx.x & 32u) ? r[0] : r[1];
}
template <typename T> __device__ __forceinline__ T sum32(T x);
template <> __device__ __forceinline__ unsigned sum32<unsigned>(unsigned x) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return r[0] | r[1];   // flags: OR
}
template <> __device__ __forceinline__ float sum32<float>(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <> __device__ __forceinline__ double sum32<double>(double x) {
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);   // (lower half's) + (upper half's): same order in both
}
template <typename T> __device__ __forceinline__ T sum16(T x);
template <> __device__ __forceinline__ unsigned sum16<unsigned>(unsigned x) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // r[0] = rows [0 0 2 2], r[1] = rows [1 1 3 3]
    return r[0] | r[1];
}
template <> __device__ __forceinline__ float sum16<float>(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <> __device__ __forceinline__ double sum16<double>(double x) {
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
template <typename T> __device__ __forceinline__ T sum8(T x);   // lanes 8 apart inside a row of 16: DPP row_ror:8
template <> __device__ __forceinline__ unsigned sum8<unsigned>(unsigned x) { return x | (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false); }
template <> __device__ __forceinline__ float sum8<float>(float x) {
    const float o = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), 0x128, 0xf, 0xf, false));
    return (threadIdx.x & 8u) ? o + x : x + o;   // (lower lane's) + (upper lane's) in both
}
template <> __device__ __forceinline__ double sum8<double>(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x128, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x128, 0xf, 0xf, false);
    const double o = __hiloint2double(hi, lo);
    return (threadIdx.x & 8u) ? o + x : x + o;
}
// total over the L lanes of a filter, identical (bit for bit) in all of them
template <int L, typename T>
__device__ __forceinline__ T sum_lanes(T x) {
    static_assert(L == 2 || L == 4 || L == 8, "lanes per filter");
    x = sum32(x);
    if constexpr (L >= 4) x = sum16(x);
    if constexpr (L >= 8) x = sum8(x);
    return x;
}

// Two values at a time: one exchange hands a's partner value to the lower lanes and b's to the upper lanes (v_permlane*_swap moves
// both directions at once), one add forms both totals, a second exchange spreads them: 7 instructions per stage for two fp64
// values where two single sums take 10.  Same additions in the same order as sum_lanes, so the same bits.
__device__ __forceinline__ void pair32(double &a, double &b) {
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    const double t = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);   // [a.lo + a.hi | b.lo + b.hi]
    const auto tl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(t), (unsigned)__double2loint(t), false, false);
    const auto th = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(t), (unsigned)__double2hiint(t), false, false);
    a = __hiloint2double((int)th[0], (int)tl[0]);
    b = __hiloint2double((int)th[1], (int)tl[1]);
}
__device__ __forceinline__ void pair16(double &a, double &b) {
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    const double t = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);   // rows [a0 + a1, b0 + b1, a2 + a3, b2 + b3]
    const auto tl = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(t), (unsigned)__double2loint(t), false, false);
    const auto th = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(t), (unsigned)__double2hiint(t), false, false);
    a = __hiloint2double((int)th[0], (int)tl[0]);
    b = __hiloint2double((int)th[1], (int)tl[1]);
}
template <int L, typename T>
__device__ __forceinline__ void sum_lanes2(T &a, T &b) {
    if constexpr (std::is_same<T, double>::value && L == 4) {
        pair32(a, b);
        pair16(a, b);
    } else {
        a = sum_lanes<L>(a);
        b = sum_lanes<L>(b);
    }
}

// LDS accesses of one phase become visible to the other lanes of the wave: DS operations of a wave execute in order, so all that
// is needed is that the COMPILER keeps the stores in front of the loads (per thread they go to different addresses)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// mat64.Dense.Inverse of the p x p innovation covariance (kb_device.h inverse_lu: LU with partial pivoting, then the columns of the
// inverse by substitution; error = exact zero pivot or |A|_inf |A^-1|_inf > 1e16) in two halves, so that the inverse never has to
// exist as a whole next to the factors: lu_factor_any keeps L and U in place and records the row exchanges (they only run when
// some lane of the wave pivots in that column: a wave-uniform test), lu_inverse_column returns one column of A^-1.
template <typename T, int P>
__device__ __forceinline__ bool lu_factor_any(T (&a)[P * P], unsigned &swaps, T &anorm, int nreal) {
    anorm = T(0);
#pragma unroll
    for (int i = 0; i < P; i++) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < P; j++) s += fabs(a[i * P + j]);
        if (i < nreal) anorm = (s > anorm || s != s) ? s : anorm;
    }
    bool bad = false;
    swaps = 0u;
    int bit = 0;
#pragma unroll
    for (int j = 0; j < P; j++) {
        bool need = false;
#pragma unroll
        for (int r = j + 1; r < P; r++) need = need || fabs(a[r * P + j]) > fabs(a[j * P + j]);
        if (__any(need)) {
#pragma unroll
            for (int r = j + 1; r < P; r++) {
                const bool sw = fabs(a[r * P + j]) > fabs(a[j * P + j]);
                swaps |= sw ? (1u << (bit + r - j - 1)) : 0u;
#pragma unroll
                for (int c = 0; c < P; c++) {   // whole rows: the multipliers move with their row (dlaswp)
                    const T t0 = a[j * P + c], t1 = a[r * P + c];
                    a[j * P + c] = sw ? t1 : t0;
                    a[r * P + c] = sw ? t0 : t1;
                }
            }
        }
        bit += P - 1 - j;
        const T piv = a[j * P + j];
        bad = bad || (piv == T(0));
        const T rp = recip(piv);   // (kb_device.h: within an ulp of 1 / piv, a third of the instructions)
        a[j * P + j] = rp;   // the solves multiply by the reciprocal (as inverse_lu does: s * (1 / a_ii))
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
// column C of the inverse: the exchanges and the elimination replayed on e_C, then the back substitution
template <typename T, int P, int C>
__device__ __forceinline__ void lu_inverse_column(const T (&a)[P * P], unsigned swaps, T (&v)[P]) {
#pragma unroll
    for (int i = 0; i < P; i++) v[i] = i == C ? T(1) : T(0);
    if (__any(swaps != 0u)) {
        int bit = 0;
#pragma unroll
        for (int j = 0; j < P; j++) {
#pragma unroll
            for (int r = j + 1; r < P; r++) {
                const bool sw = (swaps >> (bit + r - j - 1)) & 1u;
                const T t0 = v[j], t1 = v[r];
                v[j] = sw ? t1 : t0;
                v[r] = sw ? t0 : t1;
            }
            bit += P - 1 - j;
        }
#pragma unroll
        for (int j = 0; j < P; j++)
#pragma unroll
            for (int r = j + 1; r < P; r++) v[r] -= a[r * P + j] * v[j];
    } else {
        // nobody in the wave pivoted: e_C is still e_C, rows above C stay zero through the elimination (the same operations on the
        // non-zero part; the skipped ones would subtract exact zeros)
#pragma unroll
        for (int j = C; j < P; j++)
#pragma unroll
            for (int r = j + 1; r < P; r++) v[r] -= a[r * P + j] * v[j];
    }
#pragma unroll
    for (int i = P - 1; i >= 0; i--) {
        T s = v[i];
#pragma unroll
        for (int k = i + 1; k < P; k++) s -= a[i * P + k] * v[k];
        v[i] = s * a[i * P + i];
    }
}

// Reduce-scatter over the L lanes of a filter: v[i] holds this lane's PART of the i-th sum, i = q' + L r; afterwards lane q holds the
// totals of ITS indices q + L r in own[r].  Each stage halves the live values: one v_permlane*_swap hands the pair member a lane does
// not keep to its partner and receives the partner's part of the one it keeps (pair32 / pair16, first half):
// 50 instructions for 16 fp64 values over 8 lanes where 16 all-lane sums take 240.
__device__ __forceinline__ double rs32(double a, double b) {   // lanes < 32 get a's total over the two halves, lanes >= 32 b's
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double rs16(double a, double b) {   // even rows of 16 lanes get a's total over the row pair, odd rows b's
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double rs8(double a, double b) {    // lanes with bit 3 clear get a's total over the lane pair 8 apart, the others b's
    const bool up = (threadIdx.x & 8u) != 0u;
    const double send = up ? a : b, keep = up ? b : a;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(send), 0x128, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(send), 0x128, 0xf, 0xf, false);
    return keep + __hiloint2double(hi, lo);
}
__device__ __forceinline__ float rs32(float a, float b) {
    const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(x[0]) + __uint_as_float(x[1]);
}
__device__ __forceinline__ float rs16(float a, float b) {
    const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(x[0]) + __uint_as_float(x[1]);
}
__device__ __forceinline__ float rs8(float a, float b) {
    const bool up = (threadIdx.x & 8u) != 0u;
    const float send = up ? a : b, keep = up ? b : a;
    return keep + __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(send), 0x128, 0xf, 0xf, false));
}
template <int L, int NS, typename T>
__device__ __forceinline__ void reduce_scatter(const T (&v)[NS], T (&own)[NS / L]) {
    static_assert(L == 4 || L == 8, "lanes per filter");
    constexpr int RP = NS / L;
    T w1[NS / 2];
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int qq = 0; qq < L / 2; qq++) w1[qq + (L / 2) * r] = rs32(v[qq + L * r], v[qq + L / 2 + L * r]);
    T w2[NS / 4];
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int qq = 0; qq < L / 4; qq++) w2[qq + (L / 4) * r] = rs16(w1[qq + (L / 2) * r], w1[qq + L / 4 + (L / 2) * r]);
    if constexpr (L == 4) {
#pragma unroll
        for (int r = 0; r < RP; r++) own[r] = w2[r];
    } else {
#pragma unroll
        for (int r = 0; r < RP; r++) own[r] = rs8(w2[2 * r], w2[2 * r + 1]);
    }
}

// The p x p inverse ONCE per filter (p = 8): lu_factor_any / lu_inverse_column above run in every lane of the filter on all p^2 values
// (64 doubles of registers, ~2600 instructions with the row exchanges).  Here lane q owns COLUMNS q, q + L, ... of S and the L lanes run
// an in-place Gauss-Jordan inversion with partial pivoting by rows: at step j the owner of column j finds the pivot row in its own
// registers (no lane reduction) and publishes the column and the row number through LDS; every lane exchanges the two rows in its
// columns, scales row j and eliminates column j from its columns; the owner's column becomes column j of the inverse (it is preset to
// e_j, so that the same statements produce it).  The columns of the result leave in the order the row exchanges imply
// (A^-1 = (Pi A)^-1 Pi: the exchanges applied to the COLUMNS in reverse order), to sb[(c NM + k) FPW] = (S^-1)[k][c].
// Error conditions as mat64.Dense.Inverse (kb_device.h inverse_lu): an exact zero pivot; the caller forms |S| |S^-1| from the returned
// |S|_1 (= |S|_inf up to rounding: S is symmetric up to rounding) and the inverse it reads back.
template <int NM, int L> constexpr bool split_dist() {
#ifdef KB_SPLIT_NODIST
    return false;
#else
    return NM == 8;
#endif
}
template <typename T, int NM, int L>
__device__ __forceinline__ bool dist_inverse(T (&A)[NM / L][NM], const int q, T *sb, const int nreal, T &anorm) {
    static_assert(NM % L == 0 && NM <= 8, "columns are dealt out cyclically; the row numbers are kept three bits each");
    constexpr int CP = NM / L, FPW = 64 / L;
    T *const cb = sb + NM * NM * FPW;   // the pivot column, its row number, then L slots for |S|_1
    T an = T(0);
#pragma unroll
    for (int tt = 0; tt < CP; tt++) {
        T s = T(0);
#pragma unroll
        for (int r = 0; r < NM; r++) s += fabs(A[tt][r]);
        an = (q + L * tt < nreal && (s > an || s != s)) ? s : an;
    }
    bool bad = false;
    unsigned perm = 0u;
    sfor<0, NM>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = J, t = j / L, oq = j % L;
        T best = fabs(A[t][j]);
        int pr = j;
#pragma unroll
        for (int r = j + 1; r < NM; r++) {
            const T v = fabs(A[t][r]);
            const bool g = v > best;   // (the first of equal maxima: idamax)
            best = g ? v : best;
            pr = g ? r : pr;
        }
        wave_lds_fence();
        if (q == oq) {
#pragma unroll
            for (int r = 0; r < NM; r++) cb[r * FPW] = A[t][r];
            cb[NM * FPW] = (T)pr;
        }
        wave_lds_fence();
        T col[NM];
#pragma unroll
        for (int r = 0; r < NM; r++) col[r] = cb[r * FPW];
        const int prj = (int)cb[NM * FPW] & 7;
        perm |= (unsigned)prj << (3 * j);
        if (__any(prj != j)) {
#pragma unroll
            for (int r = j + 1; r < NM; r++) {
                const bool sw = prj == r;
                { const T t0 = col[j], t1 = col[r]; col[j] = sw ? t1 : t0; col[r] = sw ? t0 : t1; }
#pragma unroll
                for (int tt = 0; tt < CP; tt++) { const T t0 = A[tt][j], t1 = A[tt][r]; A[tt][j] = sw ? t1 : t0; A[tt][r] = sw ? t0 : t1; }
            }
        }
        const T piv = col[j];
        bad = bad || (piv == T(0));
        const T rp = recip(piv);
        const bool own = q == oq;
#pragma unroll
        for (int r = 0; r < NM; r++) A[t][r] = own ? (r == j ? T(1) : T(0)) : A[t][r];
#pragma unroll
        for (int tt = 0; tt < CP; tt++) {
            const T sc = A[tt][j] * rp;
#pragma unroll
            for (int r = 0; r < NM; r++)
                if (r != j) A[tt][r] -= col[r] * sc;
            A[tt][j] = sc;
#pragma unroll
            for (int r = 0; r < NM; r++) pin(A[tt][r]);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    wave_lds_fence();
#pragma unroll
    for (int tt = 0; tt < CP; tt++) {
        int c = q + L * tt;
#pragma unroll
        for (int j = NM - 1; j >= 0; j--) {
            const int pj = (int)((perm >> (3 * j)) & 7u);
            c = (c == j) ? pj : (c == pj ? j : c);
        }
        T *const dst = sb + c * (NM * FPW);
#pragma unroll
        for (int r = 0; r < NM; r++) dst[r * FPW] = A[tt][r];
    }
    cb[(NM + 1 + q) * FPW] = an;
    wave_lds_fence();
    anorm = T(0);
#pragma unroll
    for (int l = 0; l < L; l++) {
        const T s = cb[(NM + 1 + l) * FPW];
        anorm = (s > anorm || s != s) ? s : anorm;
    }
    return bad;
}

template <int NS, int NM>
constexpr int split_lds_elems() { return NS * NS > tri(NS) + NM * NS ? NS * NS : tri(NS) + NM * NS; }
template <typename T, int NS, int NM, int L>
constexpr int split_waves_per_simd() { return (int)sizeof(T) * split_lds_elems<NS, NM>() * (64 / L) * 8 <= 160 * 1024 ? 2 : 1; }

#define KB_SB() __builtin_amdgcn_sched_barrier(0)
// Where the later operands are requested (A/B runs on one box, profiles/NOTES.md): H at the start of the P- loop and R, y at the start
// of the P- H^T loop (2-4 % ahead of requesting them at their first use); Q only when F has gone to LDS (earlier costs spills)
#ifndef KB_SPLIT_HJ
#define KB_SPLIT_HJ 0
#endif
#ifndef KB_SPLIT_RL
#define KB_SPLIT_RL NS   // < NS: R is requested again at that step of the A P- loop; NS: at the start of the loop behind it
#endif
#ifndef KB_SPLIT_R1LATE
#define KB_SPLIT_R1LATE 0
#endif
#ifndef KB_SPLIT_QK
#define KB_SPLIT_QK NS   // < NS: Q is requested at that step of the T = F P loop; NS: when F has gone to LDS
#endif

// GEN = false: the batch has exactly this shape, Noiseless, FULL / PREDICT as given.  GEN = true: any n <= NS, p <= NM, m <= NC
// (zero padding as in kb_vanilla_reg.h PAD: zeros, and an identity block in R).  RT = true (GEN only): FULL / PREDICT / Noise are
// taken from the launch arguments (wave-uniform branches) -- ONE instantiation per NS carries every member of the family, at one
// wave per SIMD and 100 KB of code; RT = false: Noiseless, FULL / PREDICT as given -- the padded shapes of the common case on the
// exact kernel's schedule, one instantiation per (NS, NM in {4, 6, 8}).
//
// Scheduling.  The kernel is ~4000 instructions of straight-line code, and left alone the machine scheduler gathers loads (170
// global, 600 LDS) far ahead of their uses: 2 KB of spills per lane.  So every phase is written as a PIPELINE over chunks -- the LDS
// operands of chunk c + 1 are requested, a scheduling barrier, the arithmetic of chunk c, a scheduling barrier -- and the global
// loads of a later phase are requested where the registers for them are free: F, P, x first; Q when T = F P is done; H when the first
// rows of P- are done; R, y, G, u behind that.  The other wave of the SIMD covers what latency this leaves exposed.
// HSPLIT (S^-1 once per filter, four lanes): H passes through LDS HALF at a time in the P- H^T loop, and P- H^T, H, K take turns at
// the front of the region in the Joseph form (the lane keeps its own columns of H in registers and writes them back when they are
// needed): 155 doubles per filter at 12 / 8 where P- | H | P- H^T side by side are 270 -- two waves per SIMD.
template <int NM, int L> constexpr bool split_hsplit() { return split_dist<NM, L>() && L == 4; }
template <typename T, int NS, int NM, int L, bool GEN, bool FULLT>
constexpr int split_lds_total() {
    constexpr int KP = (tri(NS) + L - 1) / L;
    if constexpr (split_hsplit<NM, L>()) {
        constexpr int SB = NM * NM + NM + 1 + L, HH = (NM / 2) * NS;
        int e = NS * NS;                                            // F
        if (tri(NS) + (SB > HH ? SB : HH) > e) e = tri(NS) + (SB > HH ? SB : HH);   // P- | half of H, then S^-1 and the pivot column
        if (NS * NM > e) e = NS * NM;                              // P- H^T, H, K in turn
        if (KP * L > e) e = KP * L;
        return e * (64 / L);
    }
    constexpr int XOFF = split_lds_elems<NS, NM>() > KP * L ? split_lds_elems<NS, NM>() : KP * L;
    return (XOFF + (NS * NM > tri(NS) ? NS * NM : 0) + (split_dist<NM, L>() ? tri(NM) : 0)) * (64 / L);   // (+ R for K R: see the Joseph form)
}
// one wave's part of one tile: filters [64 tile + (gw % L) 64 / L, ... + 64 / L), gw = L tile + part
// HYB: the HybridKF measurement update (hybrid.go:104-204; CKF or EKF by StepArgs::ekf, no SNC, no Predict()) -- the same algebra on
// Phi, Htilde, R of the model block (kb_prepare packs them there), no Q, the measurement as (real - computed), xBar = 0 for the EKF;
// the Estimate's measurement is the real observation and it carries the prefit residual (es_dobs).
template <typename T, int NS, int NM, int NC, int L, bool GEN, bool FULLT, bool PREDT, bool RT = GEN, int NOISET = 0, bool HYB = false>
__device__ __forceinline__ void vanilla_split_part(const StepArgs &a, const int64_t gw, T *lds) {
    static_assert(NS % L == 0, "rows are dealt out cyclically");
    constexpr int FPW = 64 / L, RP = NS / L, TR = tri(NS), TM = tri(NM);
    constexpr int HOFF = TR;                       // LDS element offset of H (later: K) next to the packed P-
    constexpr int KP = (TR + L - 1) / L;           // packed elements of P per lane
    constexpr int XOFF = split_lds_elems<NS, NM>() > KP * L ? split_lds_elems<NS, NM>() : KP * L;
    constexpr bool DIST = split_dist<NM, L>();   // S^-1 once per filter (dist_inverse) instead of once per lane
    constexpr int CP = DIST ? NM / L : 1;
    constexpr bool HSPLIT = split_hsplit<NM, L>();
    constexpr int NMH = HSPLIT ? NM / 2 : NM;       // rows of H in LDS at a time while P- H^T is formed
    constexpr int GOFF = HSPLIT ? 0 : (NS * NM > TR ? XOFF : 0);   // P- H^T for the Joseph form: over P-, unless its n p elements would reach into H behind it
    constexpr int HJ = HSPLIT ? 0 : HOFF;           // H and K in the Joseph form
    const int rn = GEN ? a.n : NS, rp = GEN ? a.p : NM, rm = GEN ? (a.need_ctrl ? a.m : 0) : NC;
    if constexpr (GEN) {   // (the launcher's conditions, for the optimiser: lane offsets such as q rn 64 then provably fit 32 bits -- the scalar-base form of global_load)
        __builtin_assume(rn >= 1 && rn <= NS);
        __builtin_assume(rp >= 1 && rp <= NM);
    }
    const bool full = RT ? (a.flags & KB_FLAG_FULL_ESTIMATE) != 0 : FULLT;
    const bool predict = RT ? a.predict != 0 : PREDT;
    const unsigned lane = threadIdx.x;
    const int q = (int)((lane / FPW) & (L - 1)), f = (int)(lane & (FPW - 1));
    const int64_t tile = gw / L;
    const int slot = (int)(gw % L) * FPW + f;
    if (tile * KB_TILE + (gw % L) * FPW >= a.N) return;
    const int64_t fi = tile * KB_TILE + slot;
    const bool active = fi < a.N;

    // Addressing (as in kb_vanilla_reg.h): every base is WAVE-UNIFORM (blockIdx), element offsets are folded into it (scalar adds,
    // instruction immediates), and the lane enters as ONE unsigned 32-bit element offset -- the scalar-base form of global_load,
    // no 64-bit address pair per access.  The lane offsets: `us` the filter's slot in the tile; `uq` = us + 64 q: element (e + q)
    // of the own filter (own rows of a vector, own share of a packed triangle, own columns of H); `uf` = us + 64 q n: row q of F.
    T *const st = (T *)a.state + tile * ((int64_t)KB_TILE * (rn + tri(rn)));
    const T *const mo = (const T *)a.model + tile * a.mo_ts;
    const unsigned us = (unsigned)slot;
    const unsigned um = a.mo_ts ? (unsigned)slot : 0u;                       // model block of a shared-model batch: slot 0 of tile 0
    const unsigned uq = (unsigned)slot + (unsigned)(q * KB_TILE);
    const unsigned umq = um + (unsigned)(q * KB_TILE);
    const unsigned uf = um + (unsigned)(q * rn * KB_TILE);
    T *lf = lds + f;
    // PAIRED: the regions read as CONTIGUOUS runs -- rows of F, columns of H (stored transposed), rows of P- H^T and of K --
    // keep TWO consecutive elements per lane (16 bytes): a run is read with ds_read_b128, which the LDS array serves at 256 B/clk; the pairs
    // of 8-byte reads the compiler otherwise forms (ds_read2_b64) go at 128 B/clk (MI355X_MICROARCH.md; NOTES.md: the array is busy 50-55 %
    // of the kernel).  Element e of such a region: lp[PX(e)], lp = lds + 2 f.
#ifdef KB_SPLIT_UNPAIRED   // (A/B: 14/7 587 -> 555 us, 16/8 606 -> 575, 16/4 459 -> 451, 12/6 232 -> 229, 9/3 170 -> 166 with the pairs; NOTES.md)
    constexpr bool PAIRED = false;
#else
    constexpr bool PAIRED = true;
#endif
    T *const lp = PAIRED ? lds + 2 * f : lds + f;
    auto PX = [](int e) constexpr -> int { return PAIRED ? (e >> 1) * (2 * FPW) + (e & 1) : e * FPW; };
    auto HX = [](int c, int l, int nmrow) constexpr -> int { return PAIRED ? l * nmrow + c : c * NS + l; };   // H: [l][c] when paired
    auto lrows = [&]() -> T * {   // lf + q NM FPW: row q of an n x p matrix in LDS (late_lane: formed where it is used)
        const unsigned t = late_lane();
        return lds + (PAIRED ? 2 : 1) * (t & (FPW - 1)) + ((t / FPW) & (L - 1)) * (NM * FPW);
    };
    // ep(base, rt, c): element (rt + c) of a block -- rt wave-uniform at run time, c a compile-time constant -- as (scalar anchor,
    // made opaque to the optimiser) + (immediate within +-8 elements): left alone, instruction selection adds the part of c that
    // does not fit the 13-bit immediate to the VECTOR half of the address (a 64-bit VGPR pair and a v_lshl_add_u64 per 8 elements).
    // ep(base, rt, c): element (rt + c) of a block as (opaque scalar anchor) + (immediate): kb_device.h anchored()
    typedef __attribute__((address_space(1))) T *gptr;
    auto ep = [&](const T *ubase, int rt, int c) -> gptr { return (gptr)anchored(ubase, rt, c); };
    // (model streams: non-temporal where a lane group reads whole 128-byte segments (L <= 4); with eight lanes per filter a group reads HALF
    // a line and the part next door the other half a little later -- the streaming hint lets the line leave the L2 in between and it comes
    // from memory twice (kb_srif_split.h: 1.36x the packed reads with the hint, 1.04x without), so there the default policy)
    auto ldg = [&](const T *ubase, int rt, int c, unsigned off) { if constexpr (L == 8) return *(ep(ubase, rt, c) + off); else return __builtin_nontemporal_load(ep(ubase, rt, c) + off); };
#ifndef KB_SPLIT_LATE_COND   // (A/B: 3-4 % at 9..12 states, 12 % at 12/8; issuing the FIRST burst this way as well costs more than it gains: profiles/NOTES.md)
    // GEN, the loads BEHIND the first burst (Q, H, R): issued always -- from element 0 of the same field (which every shape has) when the
    // shape does not have the element (the run-time part of the element index takes the stand-in: scalar-base form kept), the value
    // masked after -- so that the s_waitcnt counters stay exact there (behind a branch the compiler assumes the worst at the join)
    auto ldg_if = [&](bool need, const T *ubase, int field, int rt, int c, unsigned off, unsigned off_alt, bool keep = false) {   // need: wave-uniform; keep: the value is read again later (default policy)
        if constexpr (!GEN) return need ? ((L == 8 || keep) ? *(ep(ubase, rt, c) + off) : ldg(ubase, rt, c, off)) : T(0);
        else {
#ifdef KB_FENCE_POSITIVE_CONTROL   // (the fence's positive control, never in the library: the round-4 kind of bug -- a stand-in that leaves the LAST tile's block)
            const auto pe_ = ep(ubase, need ? rt : field + a.L.mo_elems - c, c) + (need ? off : off_alt);
#else
            const auto pe_ = ep(ubase, need ? rt : field - c, c) + (need ? off : off_alt);
#endif
            const T v = (L == 8 || keep) ? *pe_ : __builtin_nontemporal_load(pe_);
            return need ? v : T(0);
        }
    };
#else
    auto ldg_if = [&](bool need, const T *ubase, int, int rt, int c, unsigned off, unsigned, bool = false) { return need ? ldg(ubase, rt, c, off) : T(0); };
#endif
    auto ldst = [&](auto NT, int rt, int c, unsigned off) {   // state block, cache policy NT (kb_vanilla_reg.h)
        const gptr pe = ep(st, rt, c) + off;
        if constexpr (decltype(NT)::value) return __builtin_nontemporal_load(pe);
        else return *pe;
    };
    auto stst = [&](auto NT, int rt, int c, unsigned off, T v) {
        const gptr pe = ep(st, rt, c) + off;
        if constexpr (decltype(NT)::value) __builtin_nontemporal_store(v, pe);
        else *pe = v;
    };
    // own rows: i_r = q + L r; rowok[r]: a real row of this batch (GEN); rowany[r] (wave-uniform): some lane's row r is real
    bool rowok[RP], rowany[RP];
#pragma unroll
    for (int r = 0; r < RP; r++) { rowok[r] = !GEN || q + L * r < rn; rowany[r] = !GEN || L * r < rn; }

    // ---- phase 0: F (own rows), x, P (a packed share per lane, handed to LDS) ---------------------------------------------
    T Fo[RP][NS], x[NS];
    {
        T Pp[KP];
        if (HYB && a.ext_phi) {   // zero copy (kb_prepare_dev): element e of filter i of the caller's planar array at ext[e ld + i]
            const T *ephi = (const T *)a.ext_phi + (active ? fi : tile * KB_TILE);
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int l = 0; l < NS; l++) {
                    const T *pe_ = ephi + (int64_t)(((rowok[r] ? q : 0) + L * r) * rn + l) * a.ext_ld;
                    const T v = (rowany[r] && l < rn) ? (L == 8 ? *pe_ : __builtin_nontemporal_load(pe_)) : T(0);   // (eight lanes: half-line segments, default policy -- see ldg)
                    Fo[r][l] = rowok[r] ? v : T(0);
                }
        } else {
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int l = 0; l < NS; l++) {
                    // (a lane whose row is padding reads lane-group 0's row, which is real whenever rowany[r])
                    const T v = (rowany[r] && l < rn) ? ldg(mo, a.L.mo_F + (GEN ? L * r * rn : 0), (GEN ? 0 : L * r * NS) + l, rowok[r] ? uf : um) : T(0);
                    Fo[r][l] = rowok[r] ? v : T(0);
                }
        }
        auto load_state = [&](auto NT) {   // cache policy of the state block: kb_vanilla_reg.h
#pragma unroll
            for (int k = 0; k < KP; k++) {
                const int e = L * k + q;
                const bool okp = e < tri(rn);
                const T v = L * k < tri(rn) ? ldst(NT, rn, L * k, okp ? uq : us) : T(0);
                Pp[k] = okp ? v : T(0);
            }
#pragma unroll
            for (int l = 0; l < NS; l++) {
                x[l] = l < rn ? ldst(NT, 0, l, us) : T(0);
            }
        };
        KB_WITH_STATE_POLICY(a, load_state);
        KB_SB();
#pragma unroll
        for (int k = 0; k < KP; k++) lf[(L * k + q) * FPW] = Pp[k];
    }
    wave_lds_fence();
    KB_SB();

    // ---- phase 1: x- = F x [+ G u], T = F P (own rows), one column of P per chunk -------------------------------------------
    T xm[RP], Tm[RP][NS];
#pragma unroll
    for (int r = 0; r < RP; r++) {
        T s = T(0);
#pragma unroll
        for (int l = 0; l < NS; l++) s += Fo[r][l] * x[l];
        xm[r] = (HYB && a.ekf) ? T(0) : s;   // (EKF: the state is the deviation from a trajectory that was just rectified, hybrid.go:166-173: x+ = K y)
        pin(xm[r]);
    }
    T Pm[RP][NS];   // [r][j] for j >= L r; the other entries are never touched.  First Q, then P-
    unsigned utri[RP];   // um + 64 tri(i_r)
#pragma unroll
    for (int r = 0; r < RP; r++) utri[r] = um + (unsigned)(((q + L * r) * (q + L * r + 1) / 2) * KB_TILE);
    auto request_Q = [&]() {
        if constexpr (HYB) {   // PBar = Phi P Phi^T [+ Gamma Q Gamma^T: PreparePNT was called for this step, hybrid.go:117-123; q <= 3]
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int j = 0; j < NS; j++) Pm[r][j] = T(0);
            if (a.snc) {
                constexpr int NQ = 3;
                const int nq = a.L.nq;
                T Qs[tri(NQ)], GQ[RP][NQ];
#pragma unroll
                for (int c = 0; c < NQ; c++)
#pragma unroll
                    for (int l = 0; l <= c; l++) Qs[symi(l, c)] = (c < nq) ? ldg(mo, a.L.mo_Q, symi(l, c), um) : T(0);
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    T gam[NQ];   // Gamma[i_r][.]: the own row
#pragma unroll
                    for (int l = 0; l < NQ; l++) {
                        const T v = (rowany[r] && l < nq) ? ldg(mo, a.L.mo_G + L * r * nq, l, rowok[r] ? um + (unsigned)(q * nq * KB_TILE) : um) : T(0);
                        gam[l] = rowok[r] ? v : T(0);
                    }
#pragma unroll
                    for (int c = 0; c < NQ; c++) {
                        T sacc = T(0);
#pragma unroll
                        for (int l = 0; l < NQ; l++) sacc += gam[l] * Qs[symi(l, c)];
                        GQ[r][c] = sacc;
                    }
                }
#pragma unroll
                for (int j = 0; j < NS; j++) {
                    T gj[NQ];   // Gamma[j][.]: the same for the L lanes of the filter
#pragma unroll
                    for (int c = 0; c < NQ; c++) gj[c] = (j < rn && c < nq) ? ldg(mo, a.L.mo_G + j * nq, c, um) : T(0);
#pragma unroll
                    for (int r = 0; r < RP; r++)
                        if (L * r <= j) {
                            T sacc = T(0);
#pragma unroll
                            for (int c = 0; c < NQ; c++) sacc += GQ[r][c] * gj[c];
                            Pm[r][j] = sacc;
                        }
                }
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < RP; r++)
    #pragma unroll
            for (int j = L * r; j < NS; j++) {
                // Q[i_r][j]: packed element (i_r, j) = tri(j) + i_r right of the diagonal, (j, i_r) = tri(i_r) + j left of it
                T v;
                const bool need = rowany[r] && j < rn;
                if (j >= L * r + L - 1) v = ldg_if(need, mo, a.L.mo_Q, a.L.mo_Q, j * (j + 1) / 2 + L * r, rowok[r] ? umq : um, um);
                else v = ldg_if(need, mo, a.L.mo_Q, a.L.mo_Q, 0, !rowok[r] ? um : (j >= q + L * r ? umq + (unsigned)((j * (j + 1) / 2 + L * r) * KB_TILE) : utri[r] + (unsigned)(j * KB_TILE)), um);
                Pm[r][j] = rowok[r] ? v : T(0);
            }
    };
    {
        T col[2][NS];
#pragma unroll
        for (int l = 0; l < NS; l++) col[0][l] = lf[symi(l, 0) * FPW];
#pragma unroll
        for (int k = 0; k < NS; k++) {
            if (k == KB_SPLIT_QK) request_Q();   // half of T is done and half of the room it needs is still free: Q arrives behind phase 1
            if (k + 1 < NS) {
#pragma unroll
                for (int l = 0; l < NS; l++) col[(k + 1) & 1][l] = lf[symi(l, k + 1) * FPW];
            }
            KB_SB();
#pragma unroll
            for (int r = 0; r < RP; r++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += Fo[r][l] * col[k & 1][l];
                Tm[r][k] = s;
                pin(Tm[r][k]);   // (kb_device.h) the products stay in THIS chunk: without it instruction selection places them at their first use
            }
            KB_SB();
        }
    }

    // ---- phase 2: F goes to LDS; P- = T F^T + Q for the own rows, columns j >= L r (the upper triangle and up to L - 1 entries
    // left of the diagonal), one row of F per chunk ------------------------------------------------------------------------
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int l = 0; l < NS; l++) (lp + q * (NS * FPW))[PX(L * r * NS + l)] = Fo[r][l];
    if constexpr (NC > 0) {   // G u while F settles in LDS
        if (rm > 0) {
            const T *up = (const T *)a.u + tile * a.u_ts;
            T u[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) u[c] = (active && c < rm) ? ldnt_at(&(up + (int64_t)c * a.u_es)[us]) : T(0);
#pragma unroll
            for (int r = 0; r < RP; r++) {
                T s = T(0);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const T g = (rowany[r] && c < rm) ? ldg(mo, a.L.mo_G + L * r * rm, c, rowok[r] ? um + (unsigned)(q * rm * KB_TILE) : um) : T(0);
                    s += (rowok[r] ? g : T(0)) * u[c];
                }
                xm[r] = xm[r] + s;
            }
        }
    }
    if (KB_SPLIT_QK >= NS) request_Q();
    wave_lds_fence();
    KB_SB();
    T Hp[NM][RP];   // H[c][i_r]: the lane's own columns of H
    // (DIST: S^-1 and the pivot columns use H's place in LDS, and the own columns of H are read a SECOND time for the Joseph form -- carried
    // through the p x p inversion they are 16-24 doubles of scratch; the first read keeps the lines in the caches)
    auto request_H = [&](bool again = false) {
        if (HYB && a.ext_phi) {   // (one branch per request, not one per element)
            // (the second request forms its addresses anew, from values the optimiser cannot tie to the first one's: kept from there, the
            // 24 sign-extended element indices and the predicates of both branches wait in scratch -- 320 B per lane at 12 / 8)
            int64_t ld = a.ext_ld;
            int rn_ = rn, q_ = q;
            if (again) { asm volatile("" : "+s"(ld)); asm volatile("" : "+s"(rn_)); asm volatile("" : "+v"(q_)); }
            const T *eh = (const T *)a.ext_h + (active ? fi : tile * KB_TILE);
#pragma unroll
            for (int c = 0; c < NM; c++)
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    const T *pe_ = eh + (int64_t)(c * rn_ + (rowok[r] ? q_ : 0) + L * r) * ld;
                    const T v = (rowany[r] && c < rp) ? ((L == 8 || (DIST && !again)) ? *pe_ : __builtin_nontemporal_load(pe_)) : T(0);
                    Hp[c][r] = rowok[r] ? v : T(0);
                }
            return;
        }
#pragma unroll
        for (int c = 0; c < NM; c++)
#pragma unroll
            for (int r = 0; r < RP; r++) {
                const T v = ldg_if(rowany[r] && c < rp, mo, a.L.mo_H, a.L.mo_H + (GEN ? c * rn : 0), (GEN ? 0 : c * NS) + L * r, rowok[r] ? umq : um, um, DIST && !again);
                Hp[c][r] = rowok[r] ? v : T(0);
            }
    };
    {
        T row[2][NS];
#pragma unroll
        for (int k = 0; k < NS; k++) row[0][k] = lp[PX(0 * NS + k)];
#pragma unroll
        for (int j = 0; j < NS; j++) {
            if (j + 1 < NS) {
#pragma unroll
                for (int k = 0; k < NS; k++) row[(j + 1) & 1][k] = lp[PX((j + 1) * NS + k)];
            }
            if (j == KB_SPLIT_HJ) request_H();
            KB_SB();
#pragma unroll
            for (int r = 0; r < RP; r++)
                if (L * r <= j) {
                    T s = T(0);
#pragma unroll
                    for (int k = 0; k < NS; k++) s += Tm[r][k] * row[j & 1][k];
                    Pm[r][j] = s + Pm[r][j];
                    pin(Pm[r][j]);
                }
            KB_SB();
        }
    }

    // ---- Noise (RT, or NOISET at compile time: 1 AWGN, 2 BatchNoise; noise.go:67-164): Process(k) into x- (vanilla.go:146), Measurement(k) into yhat (:157), Process(k) again
    // into x+ (:195), k = kf.step of THIS filter.  The normals of a draw are the filter's (kb_vanilla_reg.h draw_normals: Philox keyed
    // by the filter index), formed by each of its L lanes; the lane applies its own rows of chol(Q) (read from the model block: the
    // constructor's factor) to them.
    [[maybe_unused]] T wpost[RP], vmeas[NM];
    if constexpr (RT || NOISET) {
#pragma unroll
        for (int r = 0; r < RP; r++) wpost[r] = T(0);
#pragma unroll
        for (int c = 0; c < NM; c++) vmeas[c] = T(0);
        if (NOISET || a.noise_kind != KB_NOISE_NOISELESS) {
            const uint64_t gfi = (uint64_t)(a.first_filter + fi);
            const uint32_t stepno = (uint32_t)a.step0 - (active ? a.lag[fi] : 0u);   // kf.step of this filter
            if (NOISET == 1 || (NOISET == 0 && a.noise_kind == KB_NOISE_AWGN)) {
                // The NV standard normals of a draw are the FILTER's (Philox keyed by the filter index, two per Box-Muller block).  Its L
                // lanes share the work: lane q forms blocks q, q + L, ... -- the same instructions in every lane, a different counter --
                // and the vector is gathered through LDS (the first NV slots: F, which sat there, is consumed): a quarter / an eighth of
                // the logarithms and sines of every lane drawing all of them; the same bits.
                auto draw_coop = [&](auto NVC, uint32_t which, auto &zv) __attribute__((always_inline)) {
                    constexpr int NV = decltype(NVC)::value, NB = (NV + 1) / 2, IT = (NB + L - 1) / L;
                    wave_lds_fence();
#pragma unroll
                    for (int it = 0; it < IT; it++) {
                        const int blk = q + L * it;
                        uint32_t rr[4];
                        Philox::gen(a.seed, gfi, stepno, ((uint32_t)(a.epoch * 4 + which) << 8) | (uint32_t)blk, rr);
                        double z0, z1;
                        box_muller(rr, z0, z1);
                        if (L * it + L - 1 < NB || blk < NB) {
                            lf[(2 * blk) * FPW] = (T)z0;
                            if (2 * (L * it + L - 1) + 1 < NV || 2 * blk + 1 < NV) lf[(2 * blk + 1) * FPW] = (T)z1;
                        }
                        KB_SB();   // one Box-Muller at a time (kb_vanilla_reg.h draw_normals)
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int kk = 0; kk < NV; kk++) zv[kk] = lf[kk * FPW];
                    wave_lds_fence();
                };
                auto own_rows_of_LQ_times = [&](const T (&z)[NS], T (&w)[RP]) {
#pragma unroll
                    for (int r = 0; r < RP; r++) {
                        T sacc = T(0);
#pragma unroll
                        for (int kk = 0; kk < L * r + L; kk++) {   // L[i_r][kk], kk <= i_r, at packed element tri(i_r) + kk
                            const bool in = rowok[r] && kk <= q + L * r;
                            const T l = (rowany[r] && kk < rn) ? ldg(mo, a.L.mo_LQ, kk, in ? utri[r] : um) : T(0);
                            sacc += (in ? l : T(0)) * z[kk];
                        }
                        w[r] = sacc;
                    }
                };
                T z[NS], w[RP];
                draw_coop(std::integral_constant<int, NS>{}, 0u, z);
                own_rows_of_LQ_times(z, w);
#pragma unroll
                for (int r = 0; r < RP; r++) { xm[r] += w[r]; pin(xm[r]); }
                KB_SB();
                if (!predict) {
                    draw_coop(std::integral_constant<int, NS>{}, 2u, z);
                    own_rows_of_LQ_times(z, wpost);
#pragma unroll
                    for (int r = 0; r < RP; r++) pin(wpost[r]);
                    KB_SB();
                }
                if (full) {
                    T z1[NM];
                    draw_coop(std::integral_constant<int, NM>{}, 1u, z1);
#pragma unroll
                    for (int c = 0; c < NM; c++) {
                        T sacc = T(0);
#pragma unroll
                        for (int kk = 0; kk <= c; kk++) sacc += ((c < rp) ? ldg(mo, a.L.mo_LR, symi(kk, c), um) : T(0)) * z1[kk];
                        vmeas[c] = sacc;
                        pin(vmeas[c]);
                    }
                    KB_SB();
                }
            } else if constexpr (RT || NOISET == 2) {   // BatchNoise: the recorded vectors of step k (noise.go:72-86)
                const T *bp = (const T *)a.bn_proc + (int64_t)stepno * rn;
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    const T w = rowok[r] ? bp[q + L * r] : T(0);
                    xm[r] += w;
                    wpost[r] = predict ? T(0) : w;
                }
                if (full) {
                    const T *bm = (const T *)a.bn_meas + (int64_t)stepno * rp;
#pragma unroll
                    for (int c = 0; c < NM; c++) vmeas[c] = c < rp ? bm[c] : T(0);
                }
            }
        }
    }

    // ---- phase 3: P- (upper triangle, packed) and H go to LDS; R, y are requested ---------------------------------------------
    // R is needed twice -- in S = H P- H^T + R and in the Joseph form's K R K^T -- with the p x p factorisation in between, where 21
    // more values do not fit: it is read twice, the first time with the default cache policy (the second read finds it in the L2)
    auto load_R = [&](T (&R)[TM], auto NT) {
#pragma unroll
        for (int c = 0; c < NM; c++)
#pragma unroll
            for (int r = 0; r <= c; r++) {
                const gptr pr = ep(mo, a.L.mo_R, symi(r, c)) + um;
                R[symi(r, c)] = c < rp ? (decltype(NT)::value ? __builtin_nontemporal_load(pr) : *pr) : (r == c ? T(1) : T(0));
            }
    };
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int j = L * r; j < NS; j++) {
            if (j >= L * r + L - 1 || j >= q + L * r) lf[(j * (j + 1) / 2 + q + L * r) * FPW] = Pm[r][j];
        }
    // FULL (vanilla.go:170-179, :216-218): every extra member of the Estimate leaves where it is formed -- P- here, yhat and the
    // innovation behind the lane sums, K behind the inverse -- whether or not the step is applied in the end (a failed step has no
    // Estimate: its slots are not read); kept for one store at the end they cost the kernel half of its waves (10 KB of LDS)
    T *const es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems);
    if constexpr (RT || FULLT) {
        if (full && active) {
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int j = L * r; j < NS; j++) {
                    const gptr pe = ep(es, a.L.es_ppred, j * (j + 1) / 2 + L * r) + uq;   // (formed outside the lane-dependent branch)
                    if (rowok[r] && j < rn && (j >= L * r + L - 1 || j >= q + L * r)) __builtin_nontemporal_store(Pm[r][j], pe);
                }
        }
    }
#pragma unroll
    for (int c = 0; c < NMH; c++)
#pragma unroll
        for (int r = 0; r < RP; r++) (lp + HOFF * FPW + q * ((PAIRED ? NMH : 1) * FPW))[PX(HX(c, L * r, NMH))] = Hp[c][r];
    wave_lds_fence();
    KB_SB();
    // P- from here on: the mirrored upper triangle in LDS (what AsSymDense returns, helper.go:65-84).  Own row i_r, column l:
    // packed element (l, i_r) for l < i_r -- at lrow[r] + l -- and (i_r, l) for l >= i_r -- at lcol[l] + i_r
    const T *lrow[RP];
#pragma unroll
    for (int r = 0; r < RP; r++) { const int i = q + L * r; lrow[r] = lf + (i * (i + 1) / 2) * FPW; }
    const T *lq = lf + q * FPW;
    auto pm_own = [&](int r, int l) -> T {   // r, l compile-time after unrolling
        if (l >= L * r + L - 1) return lq[(l * (l + 1) / 2 + L * r) * FPW];
        if (l < L * r) return lrow[r][l * FPW];
        return *(l >= q + L * r ? lq + (l * (l + 1) / 2 + L * r) * FPW : lrow[r] + l * FPW);
    };
    // ---- P- H^T (own rows), two columns of P- / H per chunk ----------------------------------------------------------------
    T PHt[RP][NM];
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int c = 0; c < NM; c++) PHt[r][c] = T(0);
    T R1[DIST ? 1 : TM], y[NM];   // requested here, used behind the P- H^T loop
    [[maybe_unused]] T Rown[CP][NM];   // DIST: R[.][q + L t], the lane's own columns
    [[maybe_unused]] T yreal[HYB ? NM : 1];
    [[maybe_unused]] T xo[RP];   // x_prev[i_r] (FULL: yhat = H x_prev, vanilla.go:155-157): read a second time, not carried from the top
    auto request_Ry = [&]() {
        if constexpr (RT || FULLT) {
#pragma unroll
            for (int r = 0; r < RP; r++) {
                const T v = (full && rowany[r]) ? *(ep(st, 0, L * r) + (rowok[r] ? uq : us)) : T(0);
                xo[r] = rowok[r] ? v : T(0);
            }
        }
        if constexpr (DIST) {
#pragma unroll
            for (int tt = 0; tt < CP; tt++) {
                const int c = q + L * tt;
#pragma unroll
                for (int rr = 0; rr < NM; rr++) {
                    const bool okr = c < rp && rr < rp;
                    const int e = rr <= c ? c * (c + 1) / 2 + rr : rr * (rr + 1) / 2 + c;   // packed element (min, max)
                    const T v = *(ep(mo, a.L.mo_R, 0) + (um + (unsigned)((okr ? e : 0) * KB_TILE)));
                    Rown[tt][rr] = okr ? v : (rr == c ? T(1) : T(0));
                }
            }
        } else {
            load_R(R1, std::false_type{});
        }
        const T *yp = (const T *)a.y + tile * a.y_ts;
#pragma unroll
        for (int r = 0; r < NM; r++) y[r] = (!predict && active && r < rp) ? ldnt_at(&(yp + (int64_t)r * a.y_es)[us]) : T(0);
        if constexpr (HYB) {   // y = real - computed observation (hybrid.go:156-158); the real one is the Estimate's Measurement()
            const T *yc = (const T *)a.y2 + tile * a.y2_ts;
#pragma unroll
            for (int r = 0; r < NM; r++) {
                yreal[r] = y[r];
                y[r] = y[r] - ((!predict && active && r < rp) ? ldnt_at(&(yc + (int64_t)r * a.y2_es)[us]) : T(0));
            }
        }
    };
    if (!KB_SPLIT_R1LATE) request_Ry();
    if constexpr (!HSPLIT) {
        constexpr int CH = 2, NCH = (NS + CH - 1) / CH;
        T hb[2][CH][NM], pb[2][CH][RP];
        auto fetch = [&](int ch, int b) {
#pragma unroll
            for (int d = 0; d < CH; d++) {
                const int l = ch * CH + d;
                if (l < NS) {
#pragma unroll
                    for (int c = 0; c < NM; c++) hb[b][d][c] = (lp + HOFF * FPW)[PX(HX(c, l, NM))];
#pragma unroll
                    for (int r = 0; r < RP; r++) pb[b][d][r] = pm_own(r, l);
                }
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            if (ch + 1 < NCH) fetch(ch + 1, (ch + 1) & 1);
            KB_SB();
#pragma unroll
            for (int d = 0; d < CH; d++)
                if (ch * CH + d < NS) {
#pragma unroll
                    for (int c = 0; c < NM; c++)
#pragma unroll
                        for (int r = 0; r < RP; r++) PHt[r][c] += pb[ch & 1][d][r] * hb[ch & 1][d][c];
                }
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int c = 0; c < NM; c++) pin(PHt[r][c]);
            KB_SB();
        }
    } else
    sfor<0, NM / NMH>([&](auto HH) __attribute__((always_inline)) {
        constexpr int c0 = HH * NMH;   // rows c0 .. c0 + NMH - 1 of H are in LDS
        if constexpr (HH > 0) {
            wave_lds_fence();
#pragma unroll
            for (int c = 0; c < NMH; c++)
#pragma unroll
                for (int r = 0; r < RP; r++) (lp + HOFF * FPW + q * ((PAIRED ? NMH : 1) * FPW))[PX(HX(c, L * r, NMH))] = Hp[c0 + c][r];
            wave_lds_fence();
            KB_SB();
        }
        constexpr int CH = 2, NCH = (NS + CH - 1) / CH;
        T hb[2][CH][NMH], pb[2][CH][RP];
        auto fetch = [&](int ch, int b) {
#pragma unroll
            for (int d = 0; d < CH; d++) {
                const int l = ch * CH + d;
                if (l < NS) {
#pragma unroll
                    for (int c = 0; c < NMH; c++) hb[b][d][c] = (lp + HOFF * FPW)[PX(HX(c, l, NMH))];
#pragma unroll
                    for (int r = 0; r < RP; r++) pb[b][d][r] = pm_own(r, l);
                }
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            if (ch + 1 < NCH) fetch(ch + 1, (ch + 1) & 1);
            KB_SB();
#pragma unroll
            for (int d = 0; d < CH; d++)
                if (ch * CH + d < NS) {
#pragma unroll
                    for (int c = 0; c < NMH; c++)
#pragma unroll
                        for (int r = 0; r < RP; r++) PHt[r][c0 + c] += pb[ch & 1][d][r] * hb[ch & 1][d][c];
                }
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int c = 0; c < NMH; c++) pin(PHt[r][c0 + c]);
            KB_SB();
        }
    });
    // ---- S = H P- H^T + R (upper triangle), H x-, [H x_prev]: partial sums over the own rows, then over the L lanes --------
    T S[DIST ? 1 : NM * NM], innov[NM];
    [[maybe_unused]] T Sown[CP][NM];   // DIST: S[.][q + L t]
    {
        if (KB_SPLIT_R1LATE) { request_Ry(); KB_SB(); }
        // the partial sums: S (upper triangle) | H x- | [H x_prev], summed over the L lanes two at a time
        // (DIST: S row by row, every entry -- the reference's H P- H^T is not mirrored either -- and reduce-scattered: a lane ends up with
        // the totals of its own columns only; the triangle's slots of `part` stay unused)
        constexpr int NV = TM + NM, NVF = NV + NM;
        T part[NVF];
        if constexpr (DIST) {
#pragma unroll
            for (int c1 = 0; c1 < NM; c1++) {
                T prow[NM], orow[CP];
#pragma unroll
                for (int c2 = 0; c2 < NM; c2++) {
                    T s = T(0);
#pragma unroll
                    for (int r = 0; r < RP; r++) s += Hp[c1][r] * PHt[r][c2];
                    prow[c2] = s;
                }
                reduce_scatter<L, NM, T>(prow, orow);
#pragma unroll
                for (int tt = 0; tt < CP; tt++) { Sown[tt][c1] = orow[tt] + Rown[tt][c1]; pin(Sown[tt][c1]); }
                if (c1 & 1) KB_SB();
            }
#pragma unroll
            for (int e = 0; e < TM; e++) part[e] = T(0);
        } else {
#pragma unroll
        for (int c2 = 0; c2 < NM; c2++)
#pragma unroll
            for (int c1 = 0; c1 <= c2; c1++) {
                T s = T(0);
#pragma unroll
                for (int r = 0; r < RP; r++) s += Hp[c1][r] * PHt[r][c2];
                part[symi(c1, c2)] = s;
            }
        }
#pragma unroll
        for (int c = 0; c < NM; c++) {
            T s = T(0), s2 = T(0);
#pragma unroll
            for (int r = 0; r < RP; r++) s += Hp[c][r] * xm[r];
            if constexpr (RT || FULLT) {
                if (full) {   // (H x_prev)[c], vanilla.go:155-157
#pragma unroll
                    for (int r = 0; r < RP; r++) s2 += Hp[c][r] * xo[r];
                }
            }
            part[TM + c] = s;
            part[NV + c] = s2;
        }
#pragma unroll
        for (int e = DIST ? TM : 0; e + 1 < NV; e += 2) {
            sum_lanes2<L>(part[e], part[e + 1]);
            pin(part[e]); pin(part[e + 1]);
            if ((e & 6) == 6) KB_SB();   // a few at a time: interleaved, the chains keep all their temporaries alive
        }
        if ((NV - (DIST ? TM : 0)) & 1) part[NV - 1] = sum_lanes<L>(part[NV - 1]);
        KB_SB();
#pragma unroll
        for (int c = 0; c < NM; c++) {
            innov[c] = predict ? T(0) : y[c] - part[TM + c];   // vanilla.go:183-184
            pin(innov[c]);
        }
        if (full) {
#pragma unroll
            for (int c = 0; c < NM; c += 2) {
                if (c + 1 < NM) sum_lanes2<L>(part[NV + c], part[NV + c + 1]);
                else part[NV + c] = sum_lanes<L>(part[NV + c]);
            }
#pragma unroll
            for (int c = 0; c < NM; c++) {
                T yh = part[NV + c];
                if constexpr (RT || NOISET) yh += vmeas[c];   // Measurement(k), vanilla.go:157
                if constexpr (RT || FULLT) {
                    if (q == 0 && active && c < rp) {
                        if constexpr (HYB) {   // {innovation (0 for the EKF: hybrid.go:166-168 never forms it), real observation, prefit residual}; Predict(): zeros
                            __builtin_nontemporal_store((a.ekf || predict) ? T(0) : innov[c], ep(es, a.L.es_innov, c) + us);
                            __builtin_nontemporal_store(predict ? T(0) : yreal[c], ep(es, a.L.es_yhat, c) + us);
                            __builtin_nontemporal_store(predict ? T(0) : y[c], ep(es, a.L.es_dobs, c) + us);
                        } else {
                            __builtin_nontemporal_store(innov[c], ep(es, a.L.es_innov, c) + us);
                            __builtin_nontemporal_store(yh, ep(es, a.L.es_yhat, c) + us);
                        }
                    }
                }
            }
        }
        if constexpr (!DIST) {
#pragma unroll
        for (int c2 = 0; c2 < NM; c2++)
#pragma unroll
            for (int c1 = 0; c1 <= c2; c1++) {
                const T v = part[symi(c1, c2)] + R1[symi(c1, c2)];
                S[c1 * NM + c2] = v;
                S[c2 * NM + c1] = v;
            }
        }
    }
    KB_SB();
    // ---- K = P- H^T S^-1 (own rows), column by column of the inverse; the same bits in the L lanes of a filter ---------------
    unsigned err = 0;
    [[maybe_unused]] unsigned swaps = 0u;
    T K[RP][NM];
    {
        T anorm, inorm = T(0), rows[NM];
        T *const sb = lf + HOFF * FPW;   // DIST: H has been consumed (it returns from the registers below); S^-1 and the pivot columns take its place
        if constexpr (DIST) {
            static_assert((HOFF + NM * NM + NM + 1 + L) * FPW <= split_lds_total<T, NS, NM, L, GEN, FULLT>(), "S^-1 and the pivot column fit behind P-");
            if (dist_inverse<T, NM, L>(Sown, q, sb, rp, anorm)) err = KB_ST_SINGULAR;
        } else {
            if (lu_factor_any<T, NM>(S, swaps, anorm, rp)) err = KB_ST_SINGULAR;
        }
#pragma unroll
        for (int i = 0; i < NM; i++) rows[i] = T(0);
        sfor<0, NM>([&](auto C) __attribute__((always_inline)) {
            constexpr int c = C;
            T v[NM];
            if constexpr (DIST) {
#pragma unroll
                for (int k = 0; k < NM; k++) v[k] = sb[(c * NM + k) * FPW];
            } else {
                lu_inverse_column<T, NM, c>(S, swaps, v);
            }
#pragma unroll
            for (int i = 0; i < NM; i++) rows[i] += fabs(v[i]);
#pragma unroll
            for (int r = 0; r < RP; r++) {
                T s = T(0);
#pragma unroll
                for (int k = 0; k < NM; k++) s += PHt[r][k] * v[k];
                K[r][c] = s;
                pin(K[r][c]);
            }
#pragma unroll
            for (int i = 0; i < NM; i++) pin(rows[i]);
            KB_SB();   // one column at a time: interleaved, the six independent solves keep six sets of temporaries alive
        });
#pragma unroll
        for (int i = 0; i < NM; i++)
            if (i < rp) inorm = (rows[i] > inorm || rows[i] != rows[i]) ? rows[i] : inorm;
        if (!(anorm * inorm <= T(1e16))) err = KB_ST_SINGULAR;
        if (HYB && predict) err = 0;   // HybridKF.Predict() (hybrid.go:125-143) forms no gain: whatever S is, it cannot fail the step
    }
    if constexpr (RT || FULLT) {
        if (full && active) {
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int c = 0; c < NM; c++)
                    if (rowok[r] && c < rp) __builtin_nontemporal_store((HYB && predict) ? T(0) : K[r][c], ep(es, a.L.es_gain + L * r * a.pmax, c) + (us + (unsigned)(q * a.pmax * KB_TILE)));
        }
    }
    KB_SB();

    T xn[RP];
    T Pn[RP][NS];   // P+, own rows, columns j >= L r
    if (predict) {
        // vanilla.go:170-179: estimate = {x-, yhat, 0, sym(P-), sym(P-), K}
#pragma unroll
        for (int r = 0; r < RP; r++) {
            xn[r] = xm[r];
#pragma unroll
            for (int j = L * r; j < NS; j++) Pn[r][j] = pm_own(r, j);   // (read back: no register waits through the factorisation)
        }
    } else {
#pragma unroll
        for (int r = 0; r < RP; r++) {
            T s = T(0);
#pragma unroll
            for (int c = 0; c < NM; c++) s += K[r][c] * innov[c];
            xn[r] = xm[r] + s;
            if constexpr (RT || NOISET) xn[r] += wpost[r];   // vanilla.go:195: Process(k) a second time
        }
        // ---- Joseph form (see the header): AP = (I - K H) P- = P- - K (P- H^T)^T for the own rows.  The own rows of P- are read
        // into AP, then P- H^T -- every row of it is needed -- goes to LDS in P-'s place (behind everything else when its n p
        // elements would not fit in front of H); two columns of AP per chunk
        // (HSPLIT: R for K R passes through LDS -- each lane loads a quarter of the triangle, 9 values where the whole of it is 36 per lane,
        // and the product reads it column by column; its padding may be anything finite: those columns of K are exact zeros)
        constexpr int RSH = DIST ? (TM + L - 1) / L : 1, ROFF = HSPLIT ? NS * NM : XOFF + (NS * NM > TR ? NS * NM : 0);
        T AP[RP][NS], R[DIST ? 1 : TM];
        [[maybe_unused]] T Rsh[RSH];
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int k = 0; k < NS; k++) AP[r][k] = pm_own(r, k);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int c = 0; c < NM; c++) (lrows() + GOFF * FPW)[PX(L * r * NM + c)] = PHt[r][c];
        wave_lds_fence();
        KB_SB();
        if constexpr (DIST) {   // on their way while P- H^T is consumed
            request_H(true);
#pragma unroll
            for (int k = 0; k < RSH; k++) {
                const bool okp = q + L * k < tri(rp);
                const T v = L * k < tri(rp) ? ldg(mo, a.L.mo_R, L * k, okp ? umq : um) : T(0);
                Rsh[k] = okp ? v : T(0);
            }
        }
        KB_SB();
        {
            constexpr int CH = DIST ? 1 : 2, NCH = (NS + CH - 1) / CH;
            T gb[2][CH][NM];
            auto fetch = [&](int ch, int b) {
#pragma unroll
                for (int d = 0; d < CH; d++) {
                    const int k = ch * CH + d;
                    if (k < NS) {
#pragma unroll
                        for (int c = 0; c < NM; c++) gb[b][d][c] = (lp + GOFF * FPW)[PX(k * NM + c)];
                    }
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                if (ch + 1 < NCH) fetch(ch + 1, (ch + 1) & 1);
                KB_SB();
#pragma unroll
                for (int d = 0; d < CH; d++) {
                    const int k = ch * CH + d;
                    if (k < NS) {
#pragma unroll
                        for (int r = 0; r < RP; r++) {
                            T sacc = T(0);
#pragma unroll
                            for (int c = 0; c < NM; c++) sacc += K[r][c] * gb[ch & 1][d][c];
                            AP[r][k] = AP[r][k] - sacc;
                            pin(AP[r][k]);
                        }
                    }
                }
                KB_SB();
            }
        }
        // V = K R - AP H^T (own rows): W = AP H^T first, two columns of H per chunk, with R on its way again
        T V[RP][NM];
        if constexpr (DIST) {   // HSPLIT: P- H^T is consumed, H takes its place; R behind
            wave_lds_fence();
#pragma unroll
            for (int c = 0; c < NM; c++)
#pragma unroll
                for (int r = 0; r < RP; r++) (lp + HJ * FPW + q * ((PAIRED ? NM : 1) * FPW))[PX(HX(c, L * r, NM))] = Hp[c][r];
#pragma unroll
            for (int k = 0; k < RSH; k++)
                if (L * k + L - 1 < TM || q + L * k < TM) lf[(ROFF + q + L * k) * FPW] = Rsh[k];
            wave_lds_fence();
            KB_SB();
        }
        {
            constexpr int CH = DIST ? 1 : 2, NCH = (NS + CH - 1) / CH;
            T hb[2][CH][NM];
            auto fetch = [&](int ch, int b) {
#pragma unroll
                for (int d = 0; d < CH; d++)
                    if (ch * CH + d < NS) {
#pragma unroll
                        for (int c = 0; c < NM; c++) hb[b][d][c] = (lp + HJ * FPW)[PX(HX(c, ch * CH + d, NM))];
                    }
            };
            fetch(0, 0);
            if constexpr (!DIST) { if (KB_SPLIT_RL >= NS) load_R(R, std::true_type{}); }
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int c = 0; c < NM; c++) V[r][c] = T(0);
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                if (ch + 1 < NCH) fetch(ch + 1, (ch + 1) & 1);
                KB_SB();
#pragma unroll
                for (int d = 0; d < CH; d++)
                    if (ch * CH + d < NS) {
#pragma unroll
                        for (int c = 0; c < NM; c++)
#pragma unroll
                            for (int r = 0; r < RP; r++) V[r][c] -= AP[r][ch * CH + d] * hb[ch & 1][d][c];
                    }
#pragma unroll
                for (int r = 0; r < RP; r++)
#pragma unroll
                    for (int c = 0; c < NM; c++) pin(V[r][c]);
                KB_SB();
            }
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int c = 0; c < NM; c++) {
                    if constexpr (DIST) continue;
                    T s = V[r][c];
#pragma unroll
                    for (int k = 0; k < NM; k++) s += K[r][k] * R[symi(k, c)];
                    V[r][c] = s;
                    pin(V[r][c]);
                }
            if constexpr (DIST) {
#pragma unroll
                for (int c = 0; c < NM; c++) {
                    T rc[NM];
#pragma unroll
                    for (int k = 0; k < NM; k++) rc[k] = lf[(ROFF + symi(k, c)) * FPW];
#pragma unroll
                    for (int r = 0; r < RP; r++) {
                        T sacc = V[r][c];
#pragma unroll
                        for (int k = 0; k < NM; k++) sacc += K[r][k] * rc[k];
                        V[r][c] = sacc;
                        pin(V[r][c]);
                    }
                    if (c & 1) KB_SB();
                }
            }
            KB_SB();
        }
        // K takes H's place in LDS, P+ = AP + V K^T, two rows of K per chunk
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int c = 0; c < NM; c++) (lrows() + HJ * FPW)[PX(L * r * NM + c)] = K[r][c];
        wave_lds_fence();
        KB_SB();
        {
            constexpr int CH = DIST ? 1 : 2, NCH = (NS + CH - 1) / CH;
            T kb[2][CH][NM];
            auto fetch = [&](int ch, int b) {
#pragma unroll
                for (int d = 0; d < CH; d++)
                    if (ch * CH + d < NS) {
#pragma unroll
                        for (int c = 0; c < NM; c++) kb[b][d][c] = (lp + HJ * FPW)[PX((ch * CH + d) * NM + c)];
                    }
            };
            fetch(0, 0);
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                if (ch + 1 < NCH) fetch(ch + 1, (ch + 1) & 1);
                KB_SB();
#pragma unroll
                for (int d = 0; d < CH; d++) {
                    const int j = ch * CH + d;
                    if (j < NS) {
#pragma unroll
                        for (int r = 0; r < RP; r++)
                            if (L * r <= j) {
                                T s = AP[r][j];
#pragma unroll
                                for (int c = 0; c < NM; c++) s += V[r][c] * kb[ch & 1][d][c];
                                Pn[r][j] = s;
                                pin(Pn[r][j]);
                            }
                    }
                }
                KB_SB();
            }
        }
    }
    // ---- non-finite screen over the own entries (stands in for AsSymDense's NaN-failing comparison), then over the L lanes
    {
        T chk = T(0);
#pragma unroll
        for (int r = 0; r < RP; r++) {
            chk += xn[r] * T(0);
#pragma unroll
            for (int j = L * r; j < NS; j++) chk += Pn[r][j] * T(0);
        }
        err |= sum_lanes<L>((chk != chk) ? (unsigned)KB_ST_NONFINITE : 0u);
    }
    const bool ok = err == 0;
    if (active && ok) {
        auto store_state = [&](auto NT) {
#pragma unroll
            for (int r = 0; r < RP; r++) {
                if (rowok[r]) stst(NT, 0, L * r, uq, xn[r]);
#pragma unroll
                for (int j = L * r; j < NS; j++)
                    if (rowok[r] && j < rn && (j >= L * r + L - 1 || j >= q + L * r)) stst(NT, rn, j * (j + 1) / 2 + L * r, uq, Pn[r][j]);
            }
        };
        KB_WITH_STATE_POLICY(a, store_state);
    }
    // (the filter index is formed again from the lane number: kept from the top of the kernel it would occupy two registers all along)
    const unsigned lane_end = late_lane();
    if (active && err && ((lane_end / FPW) & (L - 1)) == 0)
        fail_step(a, tile * KB_TILE + (int64_t)((gw % L) * FPW + (lane_end & (FPW - 1))), err);   // vanilla.go:164-167, :207-215 return before kf.step++ (:218)
}

// Which part a workgroup takes.  With L = 8 a part is 8 filters: 64 bytes of every 512-byte element row, HALF a 128-byte line; its
// neighbour (the other half of every line) would be the next workgroup -- on the next XCD (workgroups are dealt round-robin over
// the 8 XCDs, each with its own L2): every line fetched twice (FETCH_SIZE: 11.7 KB per filter-step at 16 / 4 against 6.2 KB packed).
// So within each run of 16 workgroups, workgroups b and b + 8 (same XCD, dispatched back to back) take neighbouring parts.
template <int L>
__device__ __forceinline__ int64_t split_part_of_block(unsigned b, unsigned nblocks) {
    if constexpr (L != 8) return b;
    if (b >= (nblocks & ~15u)) return b;
    const unsigned r = b & 15u;
    return (int64_t)((b & ~15u) + ((r & 7u) << 1) + (r >> 3));
}

// One-wave workgroups (they share nothing, and a finished wave frees its slot and its LDS at once).  PERSIST: the grid is one
// workgroup per wave slot of the device and each walks over the parts gw = blockIdx, blockIdx + gridDim, ...
template <typename T, int NS, int NM, int NC, int L, bool GEN, bool FULLT, bool PREDT, bool PERSIST = false, bool RT = GEN, int NOISET = 0, bool HYB = false>
// (Hybrid at 12 / 8 on ONE wave per SIMD without scratch was measured against two waves with 132 B: 1.65 against 1.59 ms per 1M-filter step)
__global__ void __launch_bounds__(64, (((RT || (NM > 6 && !split_hsplit<NM, L>())) && L == 4) ? 1 : ((int)sizeof(T) * split_lds_total<T, NS, NM, L, RT, FULLT>() * 8 <= 160 * 1024 ? 2 : 1))) vanilla_split_kernel(const StepArgs a) {
    __shared__ __attribute__((aligned(16))) T lds[split_lds_total<T, NS, NM, L, RT, FULLT>()];
    if constexpr (PERSIST) {
        const int64_t nparts = a.ntiles * L;
        for (int64_t gw = blockIdx.x; gw < nparts; gw += gridDim.x) {
            vanilla_split_part<T, NS, NM, NC, L, GEN, FULLT, PREDT, RT, NOISET, HYB>(a, gw, lds);
            wave_lds_fence();
        }
    } else {
        vanilla_split_part<T, NS, NM, NC, L, GEN, FULLT, PREDT, RT, NOISET, HYB>(a, split_part_of_block<L>(blockIdx.x, gridDim.x), lds);
    }
}
#undef KB_SB

}  // namespace kb
