// kb_vanilla_strict.hip -- KB_FLAG_STRICT_SYMCHECK on registers: the statement-by-statement Vanilla step (vanilla.go:128-220) with
// BOTH triangles of P-, P+ and AsSymDense's tolerance test (helper.go:65-84), compile-time dimensions, every matrix in VGPRs.
//
// The flag exists for the reference's exact error behaviour: AsSymDense fails on ROUNDING-level differences between M_ij and M_ji
// (tests/test_symcheck_gpu.py), so this kernel must round exactly like gonum / the oracle / vanilla_gen_kernel: no FMA contraction
// (the pragma below covers everything this translation unit instantiates, kb_device.h included), the same operand order in every
// sum, the same pivot choice (Dgetf2: first largest entry, one row exchange).  Zero padding (PAD: run-time n <= NS, p <= NM, m <= NC)
// adds exact zeros to those sums and leaves every real entry bit-identical.  Until round 3 such batches ran the scratch-array
// generic kernel (16-30x slower at 1M filters); it remains for shapes beyond 6 / 4 / 2 and for strict batches with noise.
// One wave per SIMD: four full n x n matrices (P-, A, A P-, P+) are alive at the symmetry test.
#pragma clang fp contract(off)
#include "kb_internal.h"
#include "kb_strict.h"
#include "kb_vanilla_reg.h"

namespace kb {

template <typename T, int NS, int NM, int NC, bool FULL, bool PREDICT>
__global__ void __launch_bounds__(64, 1) vanilla_strict_kernel(const StepArgs a) {
    const int rn = a.n, rp = a.p, rm = a.m;   // real sizes: the arithmetic runs on the zero-padded operands
    const unsigned lane = threadIdx.x & 63u;
    const int64_t tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    const TilePtr<T> st{(T *)a.state + tile * ((int64_t)KB_TILE * (rn + tri(rn))), lane};
    const TilePtr<const T> mo{(const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems), lane};
    const TilePtr<const T> moF = mo.field(a.L.mo_F), moH = mo.field(a.L.mo_H), moQ = mo.field(a.L.mo_Q), moR = mo.field(a.L.mo_R),
                           moG = mo.field(a.L.mo_G);
    const T *yp = (const T *)a.y + tile * a.y_ts + lane;
    T x[NS], P[NS * NS], F[NS * NS], H[NM * NS], Q[NS * NS], R[NM * NM];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const bool in = i < rn && j < rn;
            F[i * NS + j] = in ? ldnt(moF, i * rn + j) : T(0);
            Q[i * NS + j] = in ? ldnt(moQ, symi(i, j)) : T(0);
            P[i * NS + j] = in ? st.ld(rn + symi(i, j)) : T(0);
        }
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = (i < rn) ? st.ld(i) : T(0);
#pragma unroll
    for (int r = 0; r < NM; r++) {
#pragma unroll
        for (int l = 0; l < NS; l++) H[r * NS + l] = (r < rp && l < rn) ? ldnt(moH, r * rn + l) : T(0);
#pragma unroll
        for (int c = 0; c < NM; c++) R[r * NM + c] = (r < rp && c < rp) ? ldnt(moR, symi(r, c)) : (r == c ? T(1) : T(0));
    }
    // x- = F x [+ G u]   (vanilla.go:138-146)
    T xm[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        T s = T(0);
#pragma unroll
        for (int l = 0; l < NS; l++) s += F[i * NS + l] * x[l];
        xm[i] = s;
    }
    if constexpr (NC > 0) {
        const T *up = (const T *)a.u + tile * a.u_ts + lane;
        T u[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) u[c] = (active && c < rm) ? ldnt_at(up + (int64_t)c * a.u_es) : T(0);
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int c = 0; c < NC; c++) s += ((i < rn && c < rm) ? ldnt(moG, i * rm + c) : T(0)) * u[c];
            xm[i] = xm[i] + s;
        }
    }
    // P- = F P F^T + Q, the full matrix, as the reference computes it (:149-152)
    T Pm[NS * NS];
    {
        T FP[NS * NS];
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int k = 0; k < NS; k++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += F[i * NS + l] * P[l * NS + k];
                FP[i * NS + k] = s;
            }
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) {
                T s = T(0);
#pragma unroll
                for (int k = 0; k < NS; k++) s += FP[i * NS + k] * F[j * NS + k];
                Pm[i * NS + j] = s + Q[i * NS + j];
            }
    }
    [[maybe_unused]] T yhat[NM];
    if constexpr (FULL) {   // yhat = H x_prev (:155-157)
#pragma unroll
        for (int r = 0; r < NM; r++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += H[r * NS + l] * x[l];
            yhat[r] = s;
        }
    }
    // gain (:160-168)
    T PHt[NS * NM], S[NM * NM], Si[NM * NM], K[NS * NM];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int c = 0; c < NM; c++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += Pm[i * NS + l] * H[c * NS + l];
            PHt[i * NM + c] = s;
        }
#pragma unroll
    for (int r = 0; r < NM; r++)
#pragma unroll
        for (int c = 0; c < NM; c++) {
            T s = T(0);
#pragma unroll
            for (int i = 0; i < NS; i++) s += H[r * NS + i] * PHt[i * NM + c];
            S[r * NM + c] = s + R[r * NM + c];
        }
    unsigned err = inverse_strict<T, NM>(S, Si, rp) ? KB_ST_SINGULAR : 0u;
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int c = 0; c < NM; c++) {
            T s = T(0);
#pragma unroll
            for (int k = 0; k < NM; k++) s += PHt[i * NM + k] * Si[k * NM + c];
            K[i * NM + c] = s;
        }
    T xn[NS], Pn[NS * NS];
    [[maybe_unused]] T innov[NM];
    if constexpr (PREDICT) {   // :170-179
#pragma unroll
        for (int i = 0; i < NS; i++) xn[i] = xm[i];
#pragma unroll
        for (int e = 0; e < NS * NS; e++) Pn[e] = Pm[e];
#pragma unroll
        for (int r = 0; r < NM; r++) innov[r] = T(0);
    } else {
#pragma unroll
        for (int r = 0; r < NM; r++) {
            const T yv = (active && r < rp) ? ldnt_at(yp + (int64_t)r * a.y_es) : T(0);
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += H[r * NS + l] * xm[l];
            innov[r] = yv - s;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int c = 0; c < NM; c++) s += K[i * NM + c] * innov[c];
            xn[i] = xm[i] + s;
        }
        // Joseph form (:197-205): A = I - K H; P+ = (A P-) A^T + (K R) K^T
        T A[NS * NS], AP[NS * NS], KR[NS * NM];
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
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int k = 0; k < NS; k++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += A[i * NS + l] * Pm[l * NS + k];
                AP[i * NS + k] = s;
            }
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int c = 0; c < NM; c++) {
                T s = T(0);
#pragma unroll
                for (int k = 0; k < NM; k++) s += K[i * NM + k] * ((k < rp && c < rp) ? R[k * NM + c] : T(0));   // the padding's identity block is not part of R
                KR[i * NM + c] = s;
            }
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) {
                T s = T(0), s2 = T(0);
#pragma unroll
                for (int k = 0; k < NS; k++) s += AP[i * NS + k] * A[j * NS + k];
#pragma unroll
                for (int c = 0; c < NM; c++) s2 += KR[i * NM + c] * K[j * NM + c];
                Pn[i * NS + j] = s + s2;
            }
    }
    // AsSymDense on P- and P+ (helper.go:65-84, vanilla.go:207-215); predict-only ignores the error (:173)
    bool finite = true, sym = true;
#pragma unroll
    for (int i = 0; i < NS; i++) {
        if (i < rn) finite = finite && (xn[i] * T(0) == T(0));
#pragma unroll
        for (int j = 0; j < NS; j++) {
            if (i < rn && j < rn) {
                if (j >= i) finite = finite && (Pn[i * NS + j] * T(0) == T(0));
                if (i != j && !PREDICT) {
                    sym = sym && sym_close(Pm[j * NS + i], Pm[i * NS + j]);
                    sym = sym && sym_close(Pn[j * NS + i], Pn[i * NS + j]);
                }
            }
        }
    }
    if (!finite) err |= KB_ST_NONFINITE;
    if (!sym) err |= KB_ST_ASYMMETRIC;
    if (active && !err) {
        if constexpr (FULL) {
            const TilePtr<T> es0{(T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems), lane};
            const TilePtr<T> esP = es0.field(a.L.es_ppred), esK = es0.field(a.L.es_gain), esI = es0.field(a.L.es_innov), esY = es0.field(a.L.es_yhat);
#pragma unroll
            for (int i = 0; i < NS; i++)
#pragma unroll
                for (int j = i; j < NS; j++)
                    if (j < rn) stnt(esP, symi(i, j), Pm[i * NS + j]);
#pragma unroll
            for (int i = 0; i < NS; i++)
#pragma unroll
                for (int c = 0; c < NM; c++)
                    if (i < rn && c < rp) stnt(esK, i * a.pmax + c, K[i * NM + c]);
#pragma unroll
            for (int r = 0; r < NM; r++)
                if (r < rp) { stnt(esI, r, innov[r]); stnt(esY, r, yhat[r]); }
        }
#pragma unroll
        for (int i = 0; i < NS; i++)
            if (i < rn) st.st(i, xn[i]);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = i; j < NS; j++)
                if (j < rn) st.st(rn + symi(i, j), Pn[i * NS + j]);
    }
    if (active && err) fail_step(a, fi, err);
}

template <typename T, int NS, int NM, int NC>
static bool try_strict(const Batch &b, const StepArgs &a) {
    const int m = a.need_ctrl ? a.m : 0;
    if (a.n > NS || a.p > NM || m > NC || (NC == 0) != (m == 0)) return false;
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    const dim3 grid((unsigned)a.ntiles), block(64);
#define KB_GO(FULL_, PRED_) KB_LAUNCH((vanilla_strict_kernel<T, NS, NM, NC, FULL_, PRED_>), grid, block, 0, b.stream, a)
    if (a.predict) { if (full) KB_GO(true, true); else KB_GO(false, true); }
    else           { if (full) KB_GO(true, false); else KB_GO(false, false); }
#undef KB_GO
    return true;
}

// KB_FLAG_STRICT_SYMCHECK, Noiseless, one step per launch, fp64, any n <= 6, p <= 4, m <= 2
bool launch_vanilla_strict(const Batch &b, const StepArgs &a) {
    return try_strict<double, 4, 2, 0>(b, a) || try_strict<double, 4, 2, 2>(b, a) || try_strict<double, 6, 3, 0>(b, a) || try_strict<double, 6, 4, 0>(b, a) || try_strict<double, 6, 4, 2>(b, a);
}

}  // namespace kb
