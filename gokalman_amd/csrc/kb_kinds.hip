// kb_kinds.hip -- run-time-dimension step kernels for SquareRoot, Information, SRIF and
// Hybrid filters (one filter per lane, per-lane private arrays).  They follow the reference
// statement by statement, quirks included; the register-resident specialisations for the
// benchmark shapes live in kb_squareroot_reg.hip / kb_srif_reg.hip.
//
//   squareroot.go:129-274   information.go:153-227   srif.go:101-160,298-340   hybrid.go:104-204
// The generic kernels are the statement-by-statement path (every shape, strict AsSymDense test, AWGN): no FMA contraction,
// so that sums round exactly as gonum's (and the oracle's) separate multiply and add do.  The strict symmetry test decides
// on rounding-level differences between M_ij and M_ji (helper.go:75); with fused products it would decide differently.
#pragma clang fp contract(off)
#include "kb_dense.h"
#include "kb_internal.h"

namespace kb {


template <typename T, int LD>
__device__ inline T awgn_component(const StepArgs &a, const T *mo, int off_L, int64_t gfi, uint32_t stepno, uint32_t which, int i, const T *z) {
    T s = T(0);
    for (int k = 0; k <= i; k++) s += ldt(mo, off_L + symi(k, i)) * z[k];
    (void)a; (void)gfi; (void)stepno; (void)which;
    return s;
}

// =====================================================================================
// SquareRoot (squareroot.go:129-274)
// =====================================================================================
template <typename T, int LD>
__global__ void __launch_bounds__(64) squareroot_gen_kernel(const StepArgs a) {
    constexpr int PD = 2 * LD;
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t fi = tile * KB_TILE + lane;
    if (fi >= a.N) return;
    const int n = a.n, p = a.p, m = a.m, sp = a.sqrt_p;
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * a.L.st_elems) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *yp = (const T *)a.y + tile * a.y_ts + lane;
    const T *up = a.u ? (const T *)a.u + tile * a.u_ts + lane : nullptr;

    T x[LD], S[LD * LD], F[LD * LD], H[LD * LD];
    for (int i = 0; i < n; i++) x[i] = ldt(st, a.L.st_vec + i);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            S[i * LD + j] = (j <= i) ? ldt(st, a.L.st_mat + symi(j, i)) : T(0);
            F[i * LD + j] = ldt(mo, a.L.mo_F + i * n + j);
        }
    for (int r = 0; r < p; r++)
        for (int j = 0; j < n; j++) H[r * LD + j] = ldt(mo, a.L.mo_H + r * n + j);
    unsigned err_acc = 0;
    for (int t = 0; t < a.nsteps; t++) {
        const uint32_t stepno = (uint32_t)(a.step0 + t) - a.lag[fi];   // kf.step of this filter
        // :139-147 x- = F x [+ G u]   (no process noise here)
        T xm[LD];
        mv_n<T, LD>(n, n, F, x, xm);
        if (a.need_ctrl)
            for (int i = 0; i < n; i++) {
                T s = T(0);
                for (int c = 0; c < m; c++) s += ldt(mo, a.L.mo_G + i * m + c) * up[(int64_t)t * a.u_step + (int64_t)c * a.u_es];
                xm[i] = xm[i] + s;
            }
        // :155-185 C = [S^T F^T ; sqrtQ^T], Uc = R-factor; QUIRK S- := Uc
        T C[PD * LD];
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                T s = T(0);
                for (int l = 0; l < n; l++) s += S[l * LD + i] * F[j * LD + l];
                C[i * LD + j] = s;
                C[(n + i) * LD + j] = (j >= i) ? ldt(mo, a.L.mo_LQ + symi(i, j)) : T(0);  // sqrtQ^T[i][j] = L[j][i]
            }
        qr_r_rt<T, LD>(2 * n, n, C);
        T Sm[LD * LD];
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Sm[i * LD + j] = (j >= i) ? C[i * LD + j] : T(0);
        // :190-216 Delta = [[sqrtR^T, 0],[S-^T H^T, S-^T]]
        const int d = n + p;
        T D[PD * PD];
        for (int r = 0; r < d; r++)
            for (int c = 0; c < d; c++) {
                T val;
                if (c < sp) {
                    if (r < sp) val = (c >= r) ? ldt(mo, a.L.mo_LR + symi(r, c)) : T(0);  // sqrtR^T[r][c] = L[c][r]
                    else {
                        T s = T(0);  // (S-^T H^T)[r-sp][c]
                        for (int l = 0; l < n; l++) s += Sm[l * LD + (r - sp)] * H[c * LD + l];
                        val = s;
                    }
                } else if (r < sp) val = T(0);
                else val = Sm[(c - p) * LD + (r - sp)];
                D[r * PD + c] = val;
            }
        qr_r_rt<T, PD>(d, d, D);
        // :225-234
        T Sp[LD * LD], Syy[LD * LD], W[LD * LD];
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Sp[i * LD + j] = (p + j <= p + i) ? D[(p + j) * PD + (p + i)] : T(0);
        for (int i = 0; i < p; i++)
            for (int j = 0; j < p; j++) Syy[i * LD + j] = (j <= i) ? D[j * PD + i] : T(0);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < p; j++) W[i * LD + j] = D[j * PD + (p + i)];
        // :237-239
        T yhat[LD];
        mv_n<T, LD>(p, n, H, x, yhat);
        if (a.noise_kind == KB_NOISE_AWGN) {
            T z[LD];
            for (int k = 0; k < p; k++) z[k] = (T)normal_at(a.seed, (uint64_t)(a.first_filter + fi), stepno, (uint32_t)(a.epoch * 4 + 1), k);
            for (int r = 0; r < p; r++) yhat[r] += awgn_component<T, LD>(a, mo, a.L.mo_LR, fi, stepno, 1, r, z);
        }
        // :242-252 K = W Syy^-1 (the inverse's error is never looked at)
        T SyyI[LD * LD], K[LD * LD];
        inverse_lu_rt<T, LD>(p, Syy, SyyI);
        mm_nn<T, LD, LD, LD>(n, p, p, W, SyyI, K);
        // :255-268
        T innov[LD], xn[LD];
        for (int r = 0; r < p; r++) {
            T s = T(0);
            for (int l = 0; l < n; l++) s += H[r * LD + l] * xm[l];
            innov[r] = yp[(int64_t)t * a.y_step + (int64_t)r * a.y_es] - s;
        }
        for (int i = 0; i < n; i++) {
            T s = T(0);
            for (int c = 0; c < p; c++) s += K[i * LD + c] * innov[c];
            xn[i] = xm[i] + s;
        }
        if (a.noise_kind == KB_NOISE_AWGN) {
            T z[LD];
            for (int k = 0; k < n; k++) z[k] = (T)normal_at(a.seed, (uint64_t)(a.first_filter + fi), stepno, (uint32_t)(a.epoch * 4 + 2), k);
            for (int i = 0; i < n; i++) xn[i] += awgn_component<T, LD>(a, mo, a.L.mo_LQ, fi, stepno, 2, i, z);
        }
        bool finite = true;
        for (int i = 0; i < n; i++) finite = finite && (xn[i] * T(0) == T(0));
        for (int i = 0; i < n; i++)
            for (int j = 0; j <= i; j++) finite = finite && (Sp[i * LD + j] * T(0) == T(0));
        unsigned err = finite ? 0u : KB_ST_NONFINITE;
        if (err_acc) err = 0;
        const bool ok = (err | err_acc) == 0;
        err_acc |= err;
        if (ok) {
            if (full) {
                T *es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane;
                for (int i = 0; i < n; i++)
                    for (int j = i; j < n; j++) stt(es, a.L.es_ppred + symi(i, j), Sm[i * LD + j]);
                for (int i = 0; i < n; i++)
                    for (int c = 0; c < p; c++) stt(es, a.L.es_gain + i * a.pmax + c, K[i * LD + c]);
                for (int r = 0; r < p; r++) { stt(es, a.L.es_innov + r, innov[r]); stt(es, a.L.es_yhat + r, yhat[r]); }
            }
            for (int i = 0; i < n; i++) x[i] = xn[i];
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) S[i * LD + j] = Sp[i * LD + j];
        }
    }
    for (int i = 0; i < n; i++) stt(st, a.L.st_vec + i, x[i]);
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) stt(st, a.L.st_mat + symi(j, i), S[i * LD + j]);
    if (err_acc) atomicOr(a.status + fi, err_acc);
}

// =====================================================================================
// Information (information.go:153-227)
// =====================================================================================
template <typename T, int LD>
__global__ void __launch_bounds__(64) information_gen_kernel(const StepArgs a) {
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t fi = tile * KB_TILE + lane;
    if (fi >= a.N) return;
    const int n = a.n, p = a.p, m = a.m, rp = a.rinv_p;
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * a.L.st_elems) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *yp = (const T *)a.y + tile * a.y_ts + lane;
    const T *up = a.u ? (const T *)a.u + tile * a.u_ts + lane : nullptr;

    T iv[LD], I[LD * LD], Fi[LD * LD], H[LD * LD];
    for (int i = 0; i < n; i++) iv[i] = ldt(st, a.L.st_vec + i);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            I[i * LD + j] = ldt(st, a.L.st_mat + symi(i, j));
            Fi[i * LD + j] = ldt(mo, a.L.mo_Finv + i * n + j);
        }
    for (int r = 0; r < p; r++)
        for (int j = 0; j < n; j++) H[r * LD + j] = ldt(mo, a.L.mo_H + r * n + j);
    unsigned err_acc = 0;
    for (int t = 0; t < a.nsteps; t++) {
        const uint32_t stepno = (uint32_t)(a.step0 + t) - a.lag[fi];   // kf.step of this filter
        // :163-165 zk = Finv^T (I Finv)
        T t1[LD * LD], zk[LD * LD], zq[LD * LD], zqi[LD * LD], Z[LD * LD];
        mm_nn<T, LD, LD, LD>(n, n, n, I, Fi, t1);
        mm_tn<T, LD, LD, LD>(n, n, n, Fi, t1, zk);
        // :169-174 Z = -zk (zk + Qinv)^-1
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) zq[i * LD + j] = zk[i * LD + j] + ldt(mo, a.L.mo_Qinv + symi(i, j));
        inverse_lu_rt<T, LD>(n, zq, zqi);
        mm_nn<T, LD, LD, LD>(n, n, n, zk, zqi, Z);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Z[i * LD + j] = T(-1) * Z[i * LD + j];
        // :176-185
        T im[LD], tv[LD], tv2[LD];
        mv_t<T, LD>(n, n, Fi, iv, im);
        if (a.need_ctrl) {
            for (int i = 0; i < n; i++) {
                T s = T(0);
                for (int c = 0; c < m; c++) s += ldt(mo, a.L.mo_G + i * m + c) * up[(int64_t)t * a.u_step + (int64_t)c * a.u_es];
                tv[i] = s;
            }
            mv_n<T, LD>(n, n, zk, tv, tv2);
            for (int i = 0; i < n; i++) im[i] = im[i] + tv2[i];
        }
        for (int i = 0; i < n; i++) {
            T s = T(0);
            for (int j = 0; j < n; j++) s += ((i == j ? T(1) : T(0)) + Z[i * LD + j]) * im[j];
            tv[i] = s;
        }
        for (int i = 0; i < n; i++) im[i] = tv[i];
        // :188-190 I- = zk + Z zk^T
        T Im[LD * LD];
        mm_nt<T, LD, LD, LD>(n, n, n, Z, zk, Im);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Im[i * LD + j] = zk[i * LD + j] + Im[i * LD + j];
        // :192-194 yhat = H State(prev) + v, State() = inverse(I) i or zeros (information.go:257-293)
        T yhat[LD];
        if (full) {
            T Ic[LD * LD], Pp[LD * LD], xp[LD];
            for (int i = 0; i < n * LD; i++) Ic[i] = T(0);
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) Ic[i * LD + j] = I[i * LD + j];
            const bool bad = inverse_lu_rt<T, LD>(n, Ic, Pp);
            for (int i = 0; i < n; i++) {
                T s = T(0);
                for (int j = 0; j < n; j++) s += (bad ? T(0) : Pp[(i <= j ? i : j) * LD + (i <= j ? j : i)]) * iv[j];
                xp[i] = s;
            }
            mv_n<T, LD>(p, n, H, xp, yhat);
            if (a.noise_kind == KB_NOISE_AWGN) {
                T z[LD];
                for (int k = 0; k < p; k++) z[k] = (T)normal_at(a.seed, (uint64_t)(a.first_filter + fi), stepno, (uint32_t)(a.epoch * 4 + 1), k);
                for (int r = 0; r < p; r++) yhat[r] += awgn_component<T, LD>(a, mo, a.L.mo_LR, fi, stepno, 1, r, z);
            }
        }
        // :197-203 HTR = H^T Rinv; QUIRK: a stale 1x1 Rinv acts as a scalar on any p
        T HTR[LD * LD];
        if (rp == 1) {
            const T r0 = ldt(mo, a.L.mo_Rinv);
            for (int i = 0; i < n; i++)
                for (int j = 0; j < p; j++) HTR[i * LD + j] = r0 * H[j * LD + i];
        } else {
            for (int i = 0; i < n; i++)
                for (int j = 0; j < p; j++) {
                    T s = T(0);
                    for (int l = 0; l < p; l++) s += H[l * LD + i] * ldt(mo, a.L.mo_Rinv + symi(l, j));
                    HTR[i * LD + j] = s;
                }
        }
        // :205-212
        T ip[LD], Ip[LD * LD];
        for (int i = 0; i < n; i++) {
            T s = T(0);
            for (int j = 0; j < p; j++) s += HTR[i * LD + j] * yp[(int64_t)t * a.y_step + (int64_t)j * a.y_es];
            ip[i] = s + im[i];
        }
        mm_nn<T, LD, LD, LD>(n, p, n, HTR, H, Ip);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Ip[i * LD + j] = Im[i * LD + j] + Ip[i * LD + j];
        // :214-222 AsSymDense on both (a panic in the reference)
        bool sym = true, finite = true;
        for (int i = 0; i < n; i++) {
            finite = finite && (ip[i] * T(0) == T(0));
            for (int j = 0; j < n; j++) {
                finite = finite && (Ip[i * LD + j] * T(0) == T(0));
                if (i != j) {
                    sym = sym && sym_close(Im[j * LD + i], Im[i * LD + j]);
                    sym = sym && sym_close(Ip[j * LD + i], Ip[i * LD + j]);
                }
            }
        }
        unsigned err = (finite ? 0u : KB_ST_NONFINITE) | ((sym || !finite) ? 0u : KB_ST_ASYMMETRIC);
        if (err_acc) err = 0;
        const bool ok = (err | err_acc) == 0;
        err_acc |= err;
        if (ok) {
            if (full) {
                T *es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane;
                for (int i = 0; i < n; i++)
                    for (int j = i; j < n; j++) stt(es, a.L.es_ppred + symi(i, j), Im[i * LD + j]);
                for (int r = 0; r < p; r++) stt(es, a.L.es_yhat + r, yhat[r]);
            }
            for (int i = 0; i < n; i++) iv[i] = ip[i];
            for (int i = 0; i < n; i++)
                for (int j = i; j < n; j++) { I[i * LD + j] = Ip[i * LD + j]; I[j * LD + i] = Ip[i * LD + j]; }
        }
    }
    for (int i = 0; i < n; i++) stt(st, a.L.st_vec + i, iv[i]);
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) stt(st, a.L.st_mat + symi(i, j), I[i * LD + j]);
    if (err_acc) atomicOr(a.status + fi, err_acc);
}

// =====================================================================================
// SRIF (srif.go:101-160 fullUpdate, :298-340 measurementSRIFUpdate, helper.go:142-172)
// =====================================================================================
template <typename T, int LD>
__global__ void __launch_bounds__(64) srif_gen_kernel(const StepArgs a) {
    constexpr int PD = 2 * LD;      // rows of the Householder panel (n + p)
    constexpr int PC = LD + 1;      // its leading dimension (n + 1 columns)
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t fi = tile * KB_TILE + lane;
    if (fi >= a.N) return;
    const int n = a.n, p = a.p;
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * a.L.st_elems) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;

    T b[LD], R[LD * LD], Phi[LD * LD], Rc[LD * LD];
    for (int i = 0; i < n; i++) b[i] = ldt(st, a.L.st_vec + i);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            R[i * LD + j] = ldt(st, a.L.st_mat + i * n + j);
            Phi[i * LD + j] = ldt(mo, a.L.mo_F + i * n + j);
        }
    unsigned err = 0;
    // :111-115 RBar = R inv(Phi)
    T PhiC[LD * LD], invPhi[LD * LD], RBar[LD * LD];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { PhiC[i * LD + j] = Phi[i * LD + j]; Rc[i * LD + j] = R[i * LD + j]; }
    if (inverse_lu_rt<T, LD>(n, PhiC, invPhi)) err |= KB_ST_SINGULAR;
    mm_nn<T, LD, LD, LD>(n, n, n, R, invPhi, RBar);
    // :118-119 xBar = Phi State(prev), State() = inv(R) b (:223-234, panics when singular); bBar = RBar xBar
    T Ri[LD * LD], xprev[LD], xBar[LD], bBar[LD];
    if (inverse_lu_rt<T, LD>(n, Rc, Ri)) err |= KB_ST_SINGULAR;
    mv_n<T, LD>(n, n, Ri, b, xprev);
    mv_n<T, LD>(n, n, Phi, xprev, xBar);
    mv_n<T, LD>(n, n, RBar, xBar, bBar);
    // :121-132 the "triangularise RBar" branch only copies (no-op quirk)
    T *es = full ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane : nullptr;
    if (a.predict) {  // :134-141
        if (!err) {
            for (int i = 0; i < n; i++) stt(st, a.L.st_vec + i, bBar[i]);
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) stt(st, a.L.st_mat + i * n + j, RBar[i * LD + j]);
            if (full) {
                for (int i = 0; i < n; i++)
                    for (int j = 0; j < n; j++) stt(es, a.L.es_ppred + i * n + j, RBar[i * LD + j]);
                for (int r = 0; r < p; r++) { stt(es, a.L.es_yhat + r, T(0)); stt(es, a.L.es_dobs + r, T(0)); }
            }
        } else fail_step(a, fi, err);   // srif.go:112-114 returns before kf.step++
        return;
    }
    const T *yr = (const T *)a.y + tile * a.y_ts + lane;
    const T *yc = (const T *)a.y2 + tile * a.y2_ts + lane;
    // :143-148 y = real - computed; whiten with "sqrtInvNoise" (QUIRK srif.go:48: it is chol_L(R))
    T yv[LD], yw[LD], real[LD];
    for (int r = 0; r < p; r++) { real[r] = yr[(int64_t)r * a.y_es]; yv[r] = real[r] - yc[(int64_t)r * a.y2_es]; }
    T A[PD * PC];
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) A[i * PC + j] = RBar[i * LD + j];
        A[i * PC + n] = bBar[i];
    }
    for (int r = 0; r < p; r++) {
        for (int j = 0; j < n; j++) {
            T s = T(0);
            for (int l = 0; l <= r; l++) s += ldt(mo, a.L.mo_LR + symi(l, r)) * ldt(mo, a.L.mo_H + l * n + j);
            A[(n + r) * PC + j] = s;
        }
        T s = T(0);
        for (int l = 0; l <= r; l++) s += ldt(mo, a.L.mo_LR + symi(l, r)) * yv[l];
        yw[r] = s;
        A[(n + r) * PC + n] = s;
    }
    householder_transf_rt<T, PC, PD>(A, n, p);  // :150 -> :298-340
    bool finite = true;
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= n; j++) finite = finite && (A[i * PC + j] * T(0) == T(0));
    if (err) { fail_step(a, fi, err); return; }   // singular Phi / R: srif.go:111-114 returns before any assignment (and before kf.step++)
    // a non-finite Householder result is stored as it is (helper.go:142-172 has no guard) and flagged, as on the register paths
    if (!finite) atomicOr(a.status + fi, (unsigned)KB_ST_NONFINITE);
    for (int i = 0; i < n; i++) stt(st, a.L.st_vec + i, A[i * PC + n]);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) stt(st, a.L.st_mat + i * n + j, A[i * PC + j]);
    if (full) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) stt(es, a.L.es_ppred + i * n + j, RBar[i * LD + j]);
        for (int r = 0; r < p; r++) { stt(es, a.L.es_yhat + r, real[r]); stt(es, a.L.es_dobs + r, yw[r]); stt(es, a.L.es_innov + r, A[(n + r) * PC + n]); }
    }
}

// =====================================================================================
// Hybrid CKF/EKF (hybrid.go:104-204)
// =====================================================================================
template <typename T, int LD>
__global__ void __launch_bounds__(64) hybrid_gen_kernel(const StepArgs a) {
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t fi = tile * KB_TILE + lane;
    if (fi >= a.N) return;
    const int n = a.n, p = a.p, q = a.L.nq;
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    const bool strict = (a.flags & KB_FLAG_STRICT_SYMCHECK) != 0;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * a.L.st_elems) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    T x[LD], P[LD * LD], Phi[LD * LD], H[LD * LD], R[LD * LD];
    for (int i = 0; i < n; i++) x[i] = ldt(st, a.L.st_vec + i);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            P[i * LD + j] = ldt(st, a.L.st_mat + symi(i, j));
            Phi[i * LD + j] = ldt(mo, a.L.mo_F + i * n + j);
        }
    for (int r = 0; r < p; r++) {
        for (int j = 0; j < n; j++) H[r * LD + j] = ldt(mo, a.L.mo_H + r * n + j);
        for (int c = 0; c < p; c++) R[r * LD + c] = ldt(mo, a.L.mo_R + symi(r, c));
    }
    // :114-123 PBar = Phi P Phi^T [+ Gamma Q Gamma^T]
    T PhiP[LD * LD], PBar[LD * LD];
    mm_nn<T, LD, LD, LD>(n, n, n, Phi, P, PhiP);
    mm_nt<T, LD, LD, LD>(n, n, n, PhiP, Phi, PBar);
    if (a.snc) {
        T GQ[LD * LD];
        for (int i = 0; i < n; i++)
            for (int c = 0; c < q; c++) {
                T s = T(0);
                for (int l = 0; l < q; l++) s += ldt(mo, a.L.mo_G + i * q + l) * ldt(mo, a.L.mo_Q + symi(l, c));
                GQ[i * LD + c] = s;
            }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                T s = T(0);
                for (int c = 0; c < q; c++) s += GQ[i * LD + c] * ldt(mo, a.L.mo_G + j * q + c);
                PBar[i * LD + j] += s;
            }
    }
    unsigned err = 0;
    T *es = full ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane : nullptr;
    if (a.predict) {  // :125-143
        T xBar[LD];
        if (a.ekf) for (int i = 0; i < n; i++) xBar[i] = T(0);
        else mv_n<T, LD>(n, n, Phi, x, xBar);
        bool sym = true, finite = true;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                finite = finite && (PBar[i * LD + j] * T(0) == T(0));
                if (i != j) sym = sym && sym_close(PBar[j * LD + i], PBar[i * LD + j]);
            }
        if (!finite) err |= KB_ST_NONFINITE; else if (!sym) err |= KB_ST_ASYMMETRIC;
        if (err) { fail_step(a, fi, err); return; }   // hybrid.go:136-138 returns before kf.step++
        for (int i = 0; i < n; i++) stt(st, a.L.st_vec + i, xBar[i]);
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) stt(st, a.L.st_mat + symi(i, j), PBar[i * LD + j]);
        if (full) {
            for (int i = 0; i < n; i++)
                for (int j = i; j < n; j++) stt(es, a.L.es_ppred + symi(i, j), PBar[i * LD + j]);
            for (int i = 0; i < n; i++)
                for (int c = 0; c < p; c++) stt(es, a.L.es_gain + i * a.pmax + c, T(0));
            for (int r = 0; r < p; r++) { stt(es, a.L.es_innov + r, T(0)); stt(es, a.L.es_yhat + r, T(0)); stt(es, a.L.es_dobs + r, T(0)); }
        }
        return;
    }
    const T *yr = (const T *)a.y + tile * a.y_ts + lane;
    const T *yc = (const T *)a.y2 + tile * a.y2_ts + lane;
    // :146-153 gain
    T PHt[LD * LD], S[LD * LD], Si[LD * LD], K[LD * LD];
    mm_nt<T, LD, LD, LD>(n, n, p, PBar, H, PHt);
    mm_nn<T, LD, LD, LD>(p, n, p, H, PHt, S);
    for (int r = 0; r < p; r++)
        for (int c = 0; c < p; c++) S[r * LD + c] += R[r * LD + c];
    if (inverse_lu_rt<T, LD>(p, S, Si)) err |= KB_ST_SINGULAR;
    mm_nn<T, LD, LD, LD>(n, p, p, PHt, Si, K);
    // :156-173
    T yv[LD], real[LD], innov[LD], xh[LD], tv[LD];
    for (int r = 0; r < p; r++) { real[r] = yr[(int64_t)r * a.y_es]; yv[r] = real[r] - yc[(int64_t)r * a.y2_es]; innov[r] = T(0); }
    if (a.ekf) {
        mv_n<T, LD>(n, p, K, yv, xh);
    } else {
        T xBar[LD];
        mv_n<T, LD>(n, n, Phi, x, xBar);
        mv_n<T, LD>(p, n, H, xBar, tv);
        for (int r = 0; r < p; r++) innov[r] = yv[r] - tv[r];
        mv_n<T, LD>(n, p, K, innov, tv);
        for (int i = 0; i < n; i++) xh[i] = xBar[i] + tv[i];
    }
    // :174-182 Joseph form
    T A[LD * LD], AP[LD * LD], Pn[LD * LD], KR[LD * LD];
    mm_nn<T, LD, LD, LD>(n, p, n, K, H, A);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[i * LD + j] = (i == j ? T(1) : T(0)) - A[i * LD + j];
    mm_nn<T, LD, LD, LD>(n, n, n, A, PBar, AP);
    mm_nn<T, LD, LD, LD>(n, p, p, K, R, KR);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            T s = T(0), s2 = T(0);
            for (int k = 0; k < n; k++) s += AP[i * LD + k] * A[j * LD + k];
            for (int c = 0; c < p; c++) s2 += KR[i * LD + c] * K[j * LD + c];
            Pn[i * LD + j] = s + s2;
        }
    bool finite = true, sym = true;
    for (int i = 0; i < n; i++) {
        finite = finite && (xh[i] * T(0) == T(0));
        for (int j = 0; j < n; j++) {
            finite = finite && (Pn[i * LD + j] * T(0) == T(0));
            if (strict && i != j) {
                sym = sym && sym_close(PBar[j * LD + i], PBar[i * LD + j]);
                sym = sym && sym_close(Pn[j * LD + i], Pn[i * LD + j]);
            }
        }
    }
    if (!finite) err |= KB_ST_NONFINITE; else if (!sym) err |= KB_ST_ASYMMETRIC;
    if (err) { fail_step(a, fi, err); return; }   // hybrid.go:150-152, :184-192 return before kf.step++
    for (int i = 0; i < n; i++) stt(st, a.L.st_vec + i, xh[i]);
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) stt(st, a.L.st_mat + symi(i, j), Pn[i * LD + j]);
    if (full) {
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) stt(es, a.L.es_ppred + symi(i, j), PBar[i * LD + j]);
        for (int i = 0; i < n; i++)
            for (int c = 0; c < p; c++) stt(es, a.L.es_gain + i * a.pmax + c, K[i * LD + c]);
        for (int r = 0; r < p; r++) { stt(es, a.L.es_innov + r, innov[r]); stt(es, a.L.es_yhat + r, real[r]); stt(es, a.L.es_dobs + r, yv[r]); }
    }
}

// =====================================================================================
// BatchKF normal equations (batch.go:41-61 SetNextMeasurement): Lambda += H^T R H, N += H^T R (real - computed)
// QUIRK: the weight is the measurement noise matrix R itself, not its inverse.
// =====================================================================================
template <typename T, int LD>
__global__ void __launch_bounds__(64) batch_ls_gen_kernel(const StepArgs a) {
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t fi = tile * KB_TILE + lane;
    if (fi >= a.N) return;
    const int n = a.n, p = a.p;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * a.L.st_elems) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *yr = (const T *)a.y + tile * a.y_ts + lane;
    const T *yc = (const T *)a.y2 + tile * a.y2_ts + lane;
    T H[LD * LD], R[LD * LD], HtR[LD * LD], y[LD];
    for (int r = 0; r < p; r++) {
        for (int j = 0; j < n; j++) H[r * LD + j] = ldt(mo, a.L.mo_H + r * n + j);
        for (int c = 0; c < p; c++) R[r * LD + c] = ldt(mo, a.L.mo_R + symi(r, c));
        y[r] = yr[(int64_t)r * a.y_es] - yc[(int64_t)r * a.y2_es];
    }
    mm_tn<T, LD, LD, LD>(n, p, p, H, R, HtR);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            T s = T(0);
            for (int l = 0; l < p; l++) s += HtR[i * LD + l] * H[l * LD + j];
            stt(st, a.L.st_mat + i * n + j, ldt(st, a.L.st_mat + i * n + j) + s);
        }
        T s = T(0);
        for (int l = 0; l < p; l++) s += HtR[i * LD + l] * y[l];
        stt(st, a.L.st_vec + i, ldt(st, a.L.st_vec + i) + s);
    }
}

// =====================================================================================
// dispatch
// =====================================================================================
#define KB_DISPATCH_GEN(KERNEL)                                                                              \
    do {                                                                                                     \
        const int dm = a.n > a.p ? (a.n > a.m ? a.n : a.m) : (a.p > a.m ? a.p : a.m);                        \
        const dim3 grid((unsigned)a.ntiles), block(64);                                                      \
        const HeavyScope hs(b, dm > 8);   /* LD = 16: scratch-heavy, see kb_internal.h */                     \
        if (b.dtype == KB_F64) {                                                                             \
            if (dm <= 4) KB_LAUNCH((KERNEL<double, 4>), grid, block, 0, hs.stream, a);              \
            else if (dm <= 8) KB_LAUNCH((KERNEL<double, 8>), grid, block, 0, hs.stream, a);         \
            else KB_LAUNCH((KERNEL<double, 16>), grid, block, 0, hs.stream, a);                     \
        } else {                                                                                             \
            if (dm <= 4) KB_LAUNCH((KERNEL<float, 4>), grid, block, 0, hs.stream, a);               \
            else if (dm <= 8) KB_LAUNCH((KERNEL<float, 8>), grid, block, 0, hs.stream, a);          \
            else KB_LAUNCH((KERNEL<float, 16>), grid, block, 0, hs.stream, a);                      \
        }                                                                                                    \
        KB_HIP(hipGetLastError());                                                                           \
    } while (0)

int launch_squareroot_gen(const Batch &b, const StepArgs &a) { KB_DISPATCH_GEN(squareroot_gen_kernel); return KB_OK; }
int launch_information_gen(const Batch &b, const StepArgs &a) { KB_DISPATCH_GEN(information_gen_kernel); return KB_OK; }
int launch_srif_gen(const Batch &b, const StepArgs &a) { KB_DISPATCH_GEN(srif_gen_kernel); return KB_OK; }
int launch_hybrid_gen(const Batch &b, const StepArgs &a) { KB_DISPATCH_GEN(hybrid_gen_kernel); return KB_OK; }
int launch_batch_ls(const Batch &b, const StepArgs &a) { KB_DISPATCH_GEN(batch_ls_gen_kernel); return KB_OK; }

}  // namespace kb
