// kb_hybrid_strict.hip -- KB_FLAG_STRICT_SYMCHECK HybridKF step on registers (hybrid.go:104-204 with both triangles of PBar and P+
// and AsSymDense's tolerance test, hybrid.go:183-200): the statement-by-statement kernel of kb_kinds.hip (hybrid_gen_kernel) with
// compile-time sizes, every matrix in VGPRs / AGPRs, no FMA contraction, the same operand order in every sum and the same pivots
// -- bit-identical results (tests/test_symcheck_gpu.py), 10-20x faster.  CKF / EKF, SNC (q <= 3, zero-padded: exact zeros leave
// every sum as it is) and Predict() are wave-uniform branches.  Phi / Htilde come from the model block (kb_prepare packs them
// there for strict batches: hybrid_reg_ok).  n = 6, p = 1..3, fp64; other shapes stay on hybrid_gen_kernel.
#pragma clang fp contract(off)
#include "kb_internal.h"
#include "kb_strict.h"

namespace kb {

template <typename T, int NS, int NM, bool FULL>
__global__ void __launch_bounds__(64, 1) hybrid_strict_kernel(const StepArgs a) {
    constexpr int NQ = 3;
    const unsigned lane = threadIdx.x & 63u;
    const int64_t tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * a.L.st_elems) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    [[maybe_unused]] T *es = FULL ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane : nullptr;
    T x[NS], P[NS * NS], Phi[NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            Phi[i * NS + j] = ldnt(mo, a.L.mo_F + i * NS + j);
            P[i * NS + j] = ldt(st, a.L.st_mat + symi(i, j));
        }
#pragma unroll
    for (int i = 0; i < NS; i++) x[i] = ldt(st, a.L.st_vec + i);
    // :114-123 PBar = Phi P Phi^T [+ Gamma Q Gamma^T]
    T PBar[NS * NS];
    {
        T PhiP[NS * NS];
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += Phi[i * NS + l] * P[l * NS + j];
                PhiP[i * NS + j] = s;
            }
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += PhiP[i * NS + l] * Phi[j * NS + l];
                PBar[i * NS + j] = s;
            }
    }
    if (a.snc) {   // wave-uniform
        const int q = a.L.nq;
        T Gm[NS * NQ], GQ[NS * NQ];
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int c = 0; c < NQ; c++) Gm[i * NQ + c] = (c < q) ? ldnt(mo, a.L.mo_G + i * q + c) : T(0);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int c = 0; c < NQ; c++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NQ; l++) s += Gm[i * NQ + l] * ((l < q && c < q) ? ldnt(mo, a.L.mo_Q + symi(l, c)) : T(0));
                GQ[i * NQ + c] = s;
            }
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) {
                T s = T(0);
#pragma unroll
                for (int c = 0; c < NQ; c++) s += GQ[i * NQ + c] * Gm[j * NQ + c];
                PBar[i * NS + j] += s;
            }
    }
    unsigned err = 0;
    if (a.predict) {   // :125-143, wave-uniform
        T xBar[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NS; j++) s += Phi[i * NS + j] * x[j];
            xBar[i] = a.ekf ? T(0) : s;
        }
        bool sym = true, finite = true;
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) {
                finite = finite && (PBar[i * NS + j] * T(0) == T(0));
                if (i != j) sym = sym && sym_close(PBar[j * NS + i], PBar[i * NS + j]);
            }
        if (!finite) err |= KB_ST_NONFINITE; else if (!sym) err |= KB_ST_ASYMMETRIC;
        if (active && err) fail_step(a, fi, err);   // hybrid.go:136-138 returns before kf.step++
        if (active && !err) {
#pragma unroll
            for (int i = 0; i < NS; i++) stt(st, a.L.st_vec + i, xBar[i]);
#pragma unroll
            for (int i = 0; i < NS; i++)
#pragma unroll
                for (int j = i; j < NS; j++) stt(st, a.L.st_mat + symi(i, j), PBar[i * NS + j]);
            if constexpr (FULL) {
#pragma unroll
                for (int i = 0; i < NS; i++)
#pragma unroll
                    for (int j = i; j < NS; j++) stt(es, a.L.es_ppred + symi(i, j), PBar[i * NS + j]);
#pragma unroll
                for (int i = 0; i < NS; i++)
#pragma unroll
                    for (int c = 0; c < NM; c++) stt(es, a.L.es_gain + i * a.pmax + c, T(0));
#pragma unroll
                for (int r = 0; r < NM; r++) { stt(es, a.L.es_innov + r, T(0)); stt(es, a.L.es_yhat + r, T(0)); stt(es, a.L.es_dobs + r, T(0)); }
            }
        }
        return;
    }
    T H[NM * NS], R[NM * NM];
#pragma unroll
    for (int r = 0; r < NM; r++) {
#pragma unroll
        for (int j = 0; j < NS; j++) H[r * NS + j] = ldnt(mo, a.L.mo_H + r * NS + j);
#pragma unroll
        for (int c = 0; c < NM; c++) R[r * NM + c] = ldnt(mo, a.L.mo_R + symi(r, c));
    }
    const T *yr = (const T *)a.y + tile * a.y_ts + lane;
    const T *yc = (const T *)a.y2 + tile * a.y2_ts + lane;
    // :146-153 gain
    T PHt[NS * NM], S[NM * NM], Si[NM * NM], K[NS * NM];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NM; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += PBar[i * NS + l] * H[j * NS + l];
            PHt[i * NM + j] = s;
        }
#pragma unroll
    for (int i = 0; i < NM; i++)
#pragma unroll
        for (int j = 0; j < NM; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += H[i * NS + l] * PHt[l * NM + j];
            S[i * NM + j] = s + R[i * NM + j];
        }
    if (inverse_strict<T, NM>(S, Si, NM)) err |= KB_ST_SINGULAR;
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NM; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NM; l++) s += PHt[i * NM + l] * Si[l * NM + j];
            K[i * NM + j] = s;
        }
    // :156-173
    T yv[NM], real[NM], innov[NM], xh[NS];
#pragma unroll
    for (int r = 0; r < NM; r++) {
        real[r] = active ? __builtin_nontemporal_load(yr + (int64_t)r * a.y_es) : T(0);
        yv[r] = real[r] - (active ? __builtin_nontemporal_load(yc + (int64_t)r * a.y2_es) : T(0));
        innov[r] = T(0);
    }
    if (a.ekf) {   // wave-uniform
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NM; j++) s += K[i * NM + j] * yv[j];
            xh[i] = s;
        }
    } else {
        T xBar[NS], tv[NM];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NS; j++) s += Phi[i * NS + j] * x[j];
            xBar[i] = s;
        }
#pragma unroll
        for (int r = 0; r < NM; r++) {
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NS; j++) s += H[r * NS + j] * xBar[j];
            tv[r] = s;
        }
#pragma unroll
        for (int r = 0; r < NM; r++) innov[r] = yv[r] - tv[r];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int j = 0; j < NM; j++) s += K[i * NM + j] * innov[j];
            xh[i] = xBar[i] + s;
        }
    }
    // :174-182 Joseph form
    T A[NS * NS], AP[NS * NS], Pn[NS * NS], KR[NS * NM];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NM; l++) s += K[i * NM + l] * H[l * NS + j];
            A[i * NS + j] = (i == j ? T(1) : T(0)) - s;
        }
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += A[i * NS + l] * PBar[l * NS + j];
            AP[i * NS + j] = s;
        }
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NM; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NM; l++) s += K[i * NM + l] * R[l * NM + j];
            KR[i * NM + j] = s;
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
    bool finite = true, sym = true;
#pragma unroll
    for (int i = 0; i < NS; i++) {
        finite = finite && (xh[i] * T(0) == T(0));
#pragma unroll
        for (int j = 0; j < NS; j++) {
            finite = finite && (Pn[i * NS + j] * T(0) == T(0));
            if (i != j) {
                sym = sym && sym_close(PBar[j * NS + i], PBar[i * NS + j]);
                sym = sym && sym_close(Pn[j * NS + i], Pn[i * NS + j]);
            }
        }
    }
    if (!finite) err |= KB_ST_NONFINITE; else if (!sym) err |= KB_ST_ASYMMETRIC;
    if (active && err) fail_step(a, fi, err);   // hybrid.go:150-152, :184-192 return before kf.step++
    if (active && !err) {
#pragma unroll
        for (int i = 0; i < NS; i++) stt(st, a.L.st_vec + i, xh[i]);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = i; j < NS; j++) stt(st, a.L.st_mat + symi(i, j), Pn[i * NS + j]);
        if constexpr (FULL) {
#pragma unroll
            for (int i = 0; i < NS; i++)
#pragma unroll
                for (int j = i; j < NS; j++) stt(es, a.L.es_ppred + symi(i, j), PBar[i * NS + j]);
#pragma unroll
            for (int i = 0; i < NS; i++)
#pragma unroll
                for (int c = 0; c < NM; c++) stt(es, a.L.es_gain + i * a.pmax + c, K[i * NM + c]);
#pragma unroll
            for (int r = 0; r < NM; r++) { stt(es, a.L.es_innov + r, innov[r]); stt(es, a.L.es_yhat + r, real[r]); stt(es, a.L.es_dobs + r, yv[r]); }
        }
    }
}

template <typename T, int NS, int NM>
static bool try_hybrid_strict(const Batch &b, const StepArgs &a) {
    if (a.n != NS || a.p != NM || a.pmax != NM || (a.snc && a.L.nq > 3)) return false;
    const dim3 grid((unsigned)a.ntiles), block(64);
    if (a.flags & KB_FLAG_FULL_ESTIMATE) KB_LAUNCH((hybrid_strict_kernel<T, NS, NM, true>), grid, block, 0, b.stream, a);
    else KB_LAUNCH((hybrid_strict_kernel<T, NS, NM, false>), grid, block, 0, b.stream, a);
    return true;
}

bool launch_hybrid_strict(const Batch &b, const StepArgs &a) {
    if (b.dtype != KB_F64 || a.ext_phi) return false;
    return try_hybrid_strict<double, 6, 2>(b, a) || try_hybrid_strict<double, 6, 3>(b, a) || try_hybrid_strict<double, 6, 1>(b, a);
}

}  // namespace kb
