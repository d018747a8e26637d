// kb_vanilla.hip -- Vanilla / pure-predictor Vanilla step kernels (vanilla.go:128-220).
//
// Two implementations of the same arithmetic:
//   vanilla_reg_kernel : dimensions are template parameters, every matrix lives in
//                        VGPRs, all loops fully unrolled.  HBM-bound: per filter-step it
//                        streams x, P(packed), F, H, Q(packed), R(packed), y in and x, P out.
//   vanilla_gen_kernel : run-time dimensions (n, p, m <= 16), per-lane private arrays.
//                        Covers every other shape, STRICT_SYMCHECK and AWGN noise.
// One filter per lane, one tile of 64 filters per wavefront (kb_device.h).
//
// Operation order follows the reference: x- = F x [+ G u] + w; P- = F P F^T + Q;
// yhat = H x_prev + v; K = P- H^T (H P- H^T + R)^-1 with an LU-pivoted explicit inverse;
// innov = y - H x-; x+ = x- + K innov + w'; Joseph form P+ = (I-KH) P- (I-KH)^T + K R K^T;
// "symmetrisation" = keep the upper triangle (AsSymDense, helper.go:65-84).
// A filter whose S is singular / ill-conditioned or whose result is non-finite keeps its
// previous estimate and gets a status bit, as the reference's (nil, err) return does.
#include "kb_internal.h"
#include "kb_vanilla_reg.h"

namespace kb {

// (vanilla_reg_kernel and try_reg live in kb_vanilla_reg.h)

// ---------------------------------------------------------------------------------
// generic run-time-dimension kernel (private arrays with leading dimension LD)
// ---------------------------------------------------------------------------------
template <typename T, int LD>
__global__ void __launch_bounds__(64) vanilla_gen_kernel(const StepArgs a) {
#pragma clang fp contract(off)   // statement-by-statement path: products and sums round separately, as in gonum / the oracle (see kb_kinds.hip)
    const int lane = threadIdx.x & 63;
    const int64_t tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    if (fi >= a.N) return;
    const int n = a.n, p = a.p, m = a.m;
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    const bool strict = (a.flags & KB_FLAG_STRICT_SYMCHECK) != 0;
    const bool awgn = a.noise_kind == KB_NOISE_AWGN;
    const bool bnoise = a.noise_kind == KB_NOISE_BATCH;  // BatchNoise (noise.go:67-106)
    const T *bnp = (const T *)a.bn_proc, *bnm = (const T *)a.bn_meas;

    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * a.L.st_elems) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *yp = a.y ? (const T *)a.y + tile * a.y_ts + lane : nullptr;
    const T *up = a.u ? (const T *)a.u + tile * a.u_ts + lane : nullptr;

    T x[LD], P[LD * LD], F[LD * LD], H[LD * LD], Q[LD * LD], R[LD * LD];
    for (int i = 0; i < n; i++) x[i] = ldt(st, a.L.st_vec + i);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            P[i * LD + j] = ldt(st, a.L.st_mat + symi(i, j));
            F[i * LD + j] = ldt(mo, a.L.mo_F + i * n + j);
            Q[i * LD + j] = ldt(mo, a.L.mo_Q + symi(i, j));
        }
    for (int r = 0; r < p; r++) {
        for (int j = 0; j < n; j++) H[r * LD + j] = ldt(mo, a.L.mo_H + r * n + j);
        for (int c = 0; c < p; c++) R[r * LD + c] = ldt(mo, a.L.mo_R + symi(r, c));
    }
    unsigned err_acc = 0, nfail = 0;
    const uint32_t lag0 = a.lag[fi];   // kf.step of this filter = calls - failed calls (kb_internal.h)
    for (int t = 0; t < a.nsteps; t++) {
        const uint32_t stepno = (uint32_t)(a.step0 + t) - lag0;
        // x- = F x [+ G u] + w
        T xm[LD];
        for (int i = 0; i < n; i++) {
            T s = T(0);
            for (int l = 0; l < n; l++) s += F[i * LD + l] * x[l];
            xm[i] = s;
        }
        if (a.need_ctrl) {
            for (int i = 0; i < n; i++) {
                T s = T(0);
                for (int c = 0; c < m; c++)
                    s += ldt(mo, a.L.mo_G + i * m + c) * up[(int64_t)t * a.u_step + (int64_t)c * a.u_es];
                xm[i] = xm[i] + s;
            }
        }
        if (bnoise)
            for (int i = 0; i < n; i++) xm[i] += bnp[(int64_t)stepno * n + i];
        if (awgn) {  // Noise.Process(k): w = L_Q z  (noise.go:133-136)
            T z[LD];
            for (int k = 0; k < n; k++) z[k] = (T)normal_at(a.seed, (uint64_t)(a.first_filter + fi), stepno, (uint32_t)(a.epoch * 4 + 0), k);
            for (int i = 0; i < n; i++) {
                T s = T(0);
                for (int k = 0; k <= i; k++) s += ldt(mo, a.L.mo_LQ + symi(k, i)) * z[k];
                xm[i] += s;
            }
        }
        // P- = F P F^T + Q (full matrix, as the reference computes it)
        T FP[LD * LD], Pm[LD * LD];
        for (int i = 0; i < n; i++)
            for (int k = 0; k < n; k++) {
                T s = T(0);
                for (int l = 0; l < n; l++) s += F[i * LD + l] * P[l * LD + k];
                FP[i * LD + k] = s;
            }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                T s = T(0);
                for (int k = 0; k < n; k++) s += FP[i * LD + k] * F[j * LD + k];
                Pm[i * LD + j] = s + Q[i * LD + j];
            }
        // yhat = H x_prev + v
        T yhat[LD];
        for (int r = 0; r < p; r++) {
            T s = T(0);
            for (int l = 0; l < n; l++) s += H[r * LD + l] * x[l];
            yhat[r] = s;
        }
        if (bnoise)
            for (int r = 0; r < p; r++) yhat[r] += bnm[(int64_t)stepno * p + r];
        if (awgn) {
            T z[LD];
            for (int k = 0; k < p; k++) z[k] = (T)normal_at(a.seed, (uint64_t)(a.first_filter + fi), stepno, (uint32_t)(a.epoch * 4 + 1), k);
            for (int r = 0; r < p; r++) {
                T s = T(0);
                for (int k = 0; k <= r; k++) s += ldt(mo, a.L.mo_LR + symi(k, r)) * z[k];
                yhat[r] += s;
            }
        }
        // gain
        T PHt[LD * LD], S[LD * LD], Si[LD * LD], K[LD * LD];
        for (int i = 0; i < n; i++)
            for (int c = 0; c < p; c++) {
                T s = T(0);
                for (int l = 0; l < n; l++) s += Pm[i * LD + l] * H[c * LD + l];
                PHt[i * LD + c] = s;
            }
        for (int r = 0; r < p; r++)
            for (int c = 0; c < p; c++) {
                T s = T(0);
                for (int i = 0; i < n; i++) s += H[r * LD + i] * PHt[i * LD + c];
                S[r * LD + c] = s + R[r * LD + c];
            }
        unsigned err = inverse_lu_rt<T, LD>(p, S, Si) ? KB_ST_SINGULAR : 0u;
        for (int i = 0; i < n; i++)
            for (int c = 0; c < p; c++) {
                T s = T(0);
                for (int k = 0; k < p; k++) s += PHt[i * LD + k] * Si[k * LD + c];
                K[i * LD + c] = s;
            }
        T xn[LD], Pn[LD * LD], innov[LD];
        if (a.predict) {
            for (int i = 0; i < n; i++) xn[i] = xm[i];
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) Pn[i * LD + j] = Pm[i * LD + j];
            for (int r = 0; r < p; r++) innov[r] = T(0);
        } else {
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
            if (bnoise)
                for (int i = 0; i < n; i++) xn[i] += bnp[(int64_t)stepno * n + i];
            if (awgn) {  // second Noise.Process(k) (vanilla.go:195)
                T z[LD];
                for (int k = 0; k < n; k++) z[k] = (T)normal_at(a.seed, (uint64_t)(a.first_filter + fi), stepno, (uint32_t)(a.epoch * 4 + 2), k);
                for (int i = 0; i < n; i++) {
                    T s = T(0);
                    for (int k = 0; k <= i; k++) s += ldt(mo, a.L.mo_LQ + symi(k, i)) * z[k];
                    xn[i] += s;
                }
            }
            T A[LD * LD], AP[LD * LD], KR[LD * LD];
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) {
                    T s = T(0);
                    for (int c = 0; c < p; c++) s += K[i * LD + c] * H[c * LD + j];
                    A[i * LD + j] = (i == j ? T(1) : T(0)) - s;
                }
            for (int i = 0; i < n; i++)
                for (int k = 0; k < n; k++) {
                    T s = T(0);
                    for (int l = 0; l < n; l++) s += A[i * LD + l] * Pm[l * LD + k];
                    AP[i * LD + k] = s;
                }
            for (int i = 0; i < n; i++)
                for (int c = 0; c < p; c++) {
                    T s = T(0);
                    for (int k = 0; k < p; k++) s += K[i * LD + k] * R[k * LD + c];
                    KR[i * LD + c] = s;
                }
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) {
                    T s = T(0);
                    for (int k = 0; k < n; k++) s += AP[i * LD + k] * A[j * LD + k];
                    T s2 = T(0);
                    for (int c = 0; c < p; c++) s2 += KR[i * LD + c] * K[j * LD + c];
                    Pn[i * LD + j] = s + s2;
                }
        }
        // AsSymDense on P- and P+ (helper.go:65-84): strict = the reference's tolerance test,
        // default = non-finite screen.  Predict-only ignores the error (vanilla.go:173).
        bool finite = true;
        for (int i = 0; i < n; i++) finite = finite && (xn[i] * T(0) == T(0));
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) finite = finite && (Pn[i * LD + j] * T(0) == T(0));
        if (!finite) err |= KB_ST_NONFINITE;
        if (strict && !a.predict) {
            bool sym = true;
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++)
                    if (i != j) {
                        sym = sym && sym_close(Pm[j * LD + i], Pm[i * LD + j]);
                        sym = sym && sym_close(Pn[j * LD + i], Pn[i * LD + j]);
                    }
            if (!sym) err |= KB_ST_ASYMMETRIC;
        }
        if (err_acc) err = 0;
        const bool ok = (err | err_acc) == 0;
        err_acc |= err;
        nfail += ok ? 0u : 1u;
        if (ok) {
            if (full) {
                T *es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane;
                for (int i = 0; i < n; i++)
                    for (int j = i; j < n; j++) stt(es, a.L.es_ppred + symi(i, j), Pm[i * LD + j]);
                for (int i = 0; i < n; i++)
                    for (int c = 0; c < p; c++) stt(es, a.L.es_gain + i * a.pmax + c, K[i * LD + c]);
                for (int r = 0; r < p; r++) {
                    stt(es, a.L.es_innov + r, innov[r]);
                    stt(es, a.L.es_yhat + r, yhat[r]);
                }
            }
            for (int i = 0; i < n; i++) x[i] = xn[i];
            for (int i = 0; i < n; i++)
                for (int j = i; j < n; j++) { P[i * LD + j] = Pn[i * LD + j]; P[j * LD + i] = Pn[i * LD + j]; }
        }
    }
    for (int i = 0; i < n; i++) stt(st, a.L.st_vec + i, x[i]);
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) stt(st, a.L.st_mat + symi(i, j), P[i * LD + j]);
    if (err_acc) fail_step(a, fi, err_acc, nfail);
}

// ---------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------
template <typename T>
static int launch_vanilla_t(const Batch &b, const StepArgs &a, bool fused) {
    const bool reg = !(a.flags & KB_FLAG_STATEMENT_KERNELS);
    const bool special = reg && !(a.flags & KB_FLAG_STRICT_SYMCHECK) && a.noise_kind == KB_NOISE_NOISELESS;
    bool done = false;
    if (reg && !(a.flags & KB_FLAG_STRICT_SYMCHECK) && a.mo_ts == 0 && b.dtype == KB_F64 && !fused && a.nsteps == 1)
        done = launch_vanilla_shared(b, a);   // one model for the whole batch: kb_vanilla_shared.hip
    if (!done && reg && !(a.flags & KB_FLAG_STRICT_SYMCHECK) && a.noise_kind != KB_NOISE_NOISELESS && b.dtype == KB_F64 && !fused && a.nsteps == 1)
        done = launch_vanilla_noise(b, a) || launch_vanilla_noise_padded(b, a);   // AWGN / BatchNoise on the register kernels
    if (!done && fused && a.noise_kind != KB_NOISE_NOISELESS) done = launch_vanilla_noise_fused(b, a);   // (vanilla_fused_ok said yes)
    if (reg && (a.flags & KB_FLAG_STRICT_SYMCHECK) && a.noise_kind == KB_NOISE_NOISELESS && b.dtype == KB_F64 && !fused && a.nsteps == 1)
        done = launch_vanilla_strict(b, a);
    if (special && !done) {
        done = try_reg<T, 6, 3, 0>(b, a, fused) || try_reg<T, 4, 2, 0>(b, a, fused);
        if (!done && b.dtype == KB_F64) done = launch_vanilla_extra_shapes(b, a, fused);
        if (!done && b.dtype == KB_F64 && !fused) done = launch_vanilla_padded(b, a) || launch_vanilla_padded8(b, a);
    }
    // shapes beyond the one-filter-per-lane kernels (n <= 16, p <= 8, m <= 2): one filter split over four lanes, kb_vanilla_split.h
    if (!done && reg && !(a.flags & KB_FLAG_STRICT_SYMCHECK) && b.dtype == KB_F64 && !fused && a.nsteps == 1)
        done = launch_vanilla_split12(b, a) || launch_vanilla_split16(b, a);
    if (!done) {
        const int d = a.n > a.p ? (a.n > a.m ? a.n : a.m) : (a.p > a.m ? a.p : a.m);
        const dim3 grid((unsigned)a.ntiles), block(64);
        const HeavyScope hs(b, d > 8);   // LD = 16: scratch-heavy, see kb_internal.h
        if (d <= 4) KB_LAUNCH((vanilla_gen_kernel<T, 4>), grid, block, 0, hs.stream, a);
        else if (d <= 8) KB_LAUNCH((vanilla_gen_kernel<T, 8>), grid, block, 0, hs.stream, a);
        else KB_LAUNCH((vanilla_gen_kernel<T, 16>), grid, block, 0, hs.stream, a);
    }
    KB_HIP(hipGetLastError());
    return KB_OK;
}

// Is there a time-fused register kernel (state carried in registers over a.nsteps steps) for this batch?  Only the benchmark
// shapes have one (try_reg<.., WITH_FUSED>); for everything else kb_update_steps_dev enqueues one single-step register launch
// per step (kb_api.hip) rather than dropping to the multi-step statement kernel.
bool vanilla_noise_fused_ok(const Batch &b, const StepArgs &a) {
    if (a.flags & (KB_FLAG_STRICT_SYMCHECK | KB_FLAG_STATEMENT_KERNELS | KB_FLAG_FULL_ESTIMATE)) return false;
    if (b.dtype != KB_F64 || a.predict || a.mo_ts == 0 || (a.need_ctrl ? a.m : 0) != 0) return false;
    return (a.noise_kind == KB_NOISE_AWGN || a.noise_kind == KB_NOISE_BATCH) && a.n == 6 && a.p == 3;
}

bool vanilla_fused_ok(const Batch &b, const StepArgs &a) {
    if (a.flags & (KB_FLAG_STRICT_SYMCHECK | KB_FLAG_STATEMENT_KERNELS)) return false;
    if (a.noise_kind != KB_NOISE_NOISELESS) return vanilla_noise_fused_ok(b, a);
    if ((a.need_ctrl ? a.m : 0) != 0) return false;
    return (a.n == 6 && a.p == 3) || (a.n == 4 && a.p == 2);
}

int launch_vanilla(const Batch &b, const StepArgs &a, bool fused) {
    if (b.dtype == KB_F64) return launch_vanilla_t<double>(b, a, fused);
    return launch_vanilla_t<float>(b, a, fused);
}

}  // namespace kb
