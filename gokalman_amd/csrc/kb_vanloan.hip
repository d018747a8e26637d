// kb_vanloan.hip -- VanLoan (c2d.go:13-75) for N independent continuous-time systems:
//   M = [[-A dt, Gamma W Gamma^T dt], [0, A^T dt]],  E = exp(M),  F = (E_22)^T,  Q = F E_12 (upper triangle mirrored)
// plus the Nyquist test on the eigenvalue the reference's loop ends up with (c2d.go:16-28).
//
// This is the step BEFORE the hot path (it builds per-filter F, Q from continuous models),
// run once per model, so it is written for generality, not for the roofline: one system per lane, run-time
// dimensions on private arrays (LD = 4 / 8 / 12 / 16 for n <= 2 / 4 / 6 / 8).
//   exp:   Pade-13 scaling and squaring (Higham 2005, the algorithm of mat64.Dense.Exp); the lower-order
//          approximants gonum picks for small norms differ from Pade-13 by less than their truncation bound (1e-16 rel).
//   eigen: Householder-Hessenberg + Francis double-shift QR (see oracle/vanloan_oracle.c for the unpinned part:
//          which eigenvalue ends up last).
#include <math.h>

#include "kb_dense.h"
#include "kb_internal.h"

namespace kb {

struct VlSrc {          // element e of system f at p[e*es + f*fs]   (AoS: es=1, fs=elems; planar: es=ld, fs=1; broadcast: fs=0)
    const void *p;
    int64_t es, fs;
};

template <typename T, int LD>
__device__ inline void vl_mm(int n, const T *A, const T *B, T *C) { mm_nn<T, LD, LD, LD>(n, n, n, A, B, C); }

// a <- exp(a) (n x n, leading dimension LD).  Returns true when the Pade system is singular.
template <typename T, int LD>
__device__ inline bool vl_expm(int n, T *a) {
    const T b[14] = {T(64764752532480000.), T(32382376266240000.), T(7771770303897600.), T(1187353796428800.), T(129060195264000.),
                     T(10559470521600.), T(670442572800.), T(33522128640.), T(1323241920.), T(40840800.), T(960960.), T(16380.), T(182.), T(1.)};
    T nrm = T(0);
    for (int j = 0; j < n; j++) {
        T s = T(0);
        for (int i = 0; i < n; i++) s += fabs(a[i * LD + j]);
        nrm = (s > nrm || s != s) ? s : nrm;
    }
    int s = 0;
    if (nrm > T(5.371920351148152)) {
        s = (int)ceil(log2((double)nrm / 5.371920351148152));
        s = s < 0 ? 0 : (s > 60 ? 60 : s);
        const T sc = (T)ldexp(1.0, -s);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) a[i * LD + j] *= sc;
    }
    T A2[LD * LD], A4[LD * LD], A6[LD * LD], T1[LD * LD], T2[LD * LD], U[LD * LD];
    vl_mm<T, LD>(n, a, a, A2);
    vl_mm<T, LD>(n, A2, A2, A4);
    vl_mm<T, LD>(n, A4, A2, A6);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { const int e = i * LD + j; T1[e] = b[13] * A6[e] + b[11] * A4[e] + b[9] * A2[e]; }
    vl_mm<T, LD>(n, A6, T1, T2);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { const int e = i * LD + j; T2[e] += b[7] * A6[e] + b[5] * A4[e] + b[3] * A2[e] + (i == j ? b[1] : T(0)); }
    vl_mm<T, LD>(n, a, T2, U);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { const int e = i * LD + j; T1[e] = b[12] * A6[e] + b[10] * A4[e] + b[8] * A2[e]; }
    vl_mm<T, LD>(n, A6, T1, T2);   // T2 = V (without the low-order terms yet)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            const int e = i * LD + j;
            const T v = T2[e] + b[6] * A6[e] + b[4] * A4[e] + b[2] * A2[e] + (i == j ? b[0] : T(0));
            T1[e] = v - U[e];   // (V - U) X = (V + U)
            T2[e] = v + U[e];
        }
    bool bad = false;
    for (int j = 0; j < n; j++) {
        int jp = j;
        for (int r = j + 1; r < n; r++)
            if (fabs(T1[r * LD + j]) > fabs(T1[jp * LD + j])) jp = r;
        if (jp != j)
            for (int c = 0; c < n; c++) {
                T t = T1[j * LD + c]; T1[j * LD + c] = T1[jp * LD + c]; T1[jp * LD + c] = t;
                t = T2[j * LD + c]; T2[j * LD + c] = T2[jp * LD + c]; T2[jp * LD + c] = t;
            }
        bad = bad || T1[j * LD + j] == T(0);
        for (int r = j + 1; r < n; r++) {
            const T l = T1[r * LD + j] / T1[j * LD + j];
            for (int c = j + 1; c < n; c++) T1[r * LD + c] -= l * T1[j * LD + c];
            for (int c = 0; c < n; c++) T2[r * LD + c] -= l * T2[j * LD + c];
        }
    }
    for (int i = n - 1; i >= 0; i--)
        for (int c = 0; c < n; c++) {
            T sum = T2[i * LD + c];
            for (int k = i + 1; k < n; k++) sum -= T1[i * LD + k] * T2[k * LD + c];
            T2[i * LD + c] = sum / T1[i * LD + i];
        }
    for (int k = 0; k < s; k++) {
        vl_mm<T, LD>(n, T2, T2, T1);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) T2[i * LD + j] = T1[i * LD + j];
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) a[i * LD + j] = T2[i * LD + j];
    return bad;
}

// |lambda| of the eigenvalue stored last by the Hessenberg-QR iteration; `a` (n x n, LD) is destroyed.
template <typename T, int LD>
__device__ inline T vl_last_eig_abs(int n, T *a) {
    T v[LD];
    for (int k = 0; k + 2 < n; k++) {
        T nr = T(0);
        for (int i = k + 1; i < n; i++) nr += a[i * LD + k] * a[i * LD + k];
        const T tail = nr - a[(k + 1) * LD + k] * a[(k + 1) * LD + k];
        if (tail == T(0)) continue;
        nr = sqrt(nr);
        const T alpha = a[(k + 1) * LD + k] >= T(0) ? -nr : nr;
        T vv = T(0);
        for (int i = k + 1; i < n; i++) { v[i] = a[i * LD + k]; if (i == k + 1) v[i] -= alpha; vv += v[i] * v[i]; }
        for (int j = 0; j < n; j++) {
            T s = T(0);
            for (int i = k + 1; i < n; i++) s += v[i] * a[i * LD + j];
            s = T(2) * s / vv;
            for (int i = k + 1; i < n; i++) a[i * LD + j] -= s * v[i];
        }
        for (int i = 0; i < n; i++) {
            T s = T(0);
            for (int j = k + 1; j < n; j++) s += a[i * LD + j] * v[j];
            s = T(2) * s / vv;
            for (int j = k + 1; j < n; j++) a[i * LD + j] -= s * v[j];
        }
        for (int i = k + 2; i < n; i++) a[i * LD + k] = T(0);
    }
    T anorm = T(0);
    for (int i = 0; i < n; i++)
        for (int j = (i > 0 ? i - 1 : 0); j < n; j++) anorm += fabs(a[i * LD + j]);
    // only the eigenvalue(s) that deflate first at the bottom (index n-1) are needed
    const int nn = n - 1;
    T t = T(0), p = T(0), q = T(0), r = T(0), s, x, y, z, w;
    for (int its = 0;; ) {
        int l;
        for (l = nn; l >= 1; l--) {
            s = fabs(a[(l - 1) * LD + l - 1]) + fabs(a[l * LD + l]);
            if (s == T(0)) s = anorm;
            if (fabs(a[l * LD + l - 1]) + s == s) { a[l * LD + l - 1] = T(0); break; }
        }
        x = a[nn * LD + nn];
        if (l == nn) return fabs(x + t);
        y = a[(nn - 1) * LD + nn - 1];
        w = a[nn * LD + nn - 1] * a[(nn - 1) * LD + nn];
        if (l == nn - 1) {
            p = T(0.5) * (y - x);
            q = p * p + w;
            z = sqrt(fabs(q));
            x += t;
            if (q >= T(0)) {
                z = p + (p >= T(0) ? fabs(z) : -fabs(z));
                return fabs(z != T(0) ? x - w / z : x + z);
            }
            return hypot(x + p, z);
        }
        if (its == 30 * n) return T(NAN);
        if (its % 10 == 0 && its > 0) {
            t += x;
            for (int i = 0; i <= nn; i++) a[i * LD + i] -= x;
            s = fabs(a[nn * LD + nn - 1]) + fabs(a[(nn - 1) * LD + nn - 2]);
            y = x = T(0.75) * s;
            w = T(-0.4375) * s * s;
        }
        ++its;
        int m;
        for (m = nn - 2; m >= l; m--) {
            z = a[m * LD + m];
            r = x - z; s = y - z;
            p = (r * s - w) / a[(m + 1) * LD + m] + a[m * LD + m + 1];
            q = a[(m + 1) * LD + m + 1] - z - r - s;
            r = a[(m + 2) * LD + m + 1];
            s = fabs(p) + fabs(q) + fabs(r);
            p /= s; q /= s; r /= s;
            if (m == l) break;
            const T u = fabs(a[m * LD + m - 1]) * (fabs(q) + fabs(r));
            const T vq = fabs(p) * (fabs(a[(m - 1) * LD + m - 1]) + fabs(z) + fabs(a[(m + 1) * LD + m + 1]));
            if (u + vq == vq) break;
        }
        for (int i = m + 2; i <= nn; i++) {
            a[i * LD + i - 2] = T(0);
            if (i != m + 2) a[i * LD + i - 3] = T(0);
        }
        for (int k = m; k <= nn - 1; k++) {
            if (k != m) {
                p = a[k * LD + k - 1];
                q = a[(k + 1) * LD + k - 1];
                r = (k != nn - 1) ? a[(k + 2) * LD + k - 1] : T(0);
                if ((x = fabs(p) + fabs(q) + fabs(r)) != T(0)) { p /= x; q /= x; r /= x; }
            }
            s = sqrt(p * p + q * q + r * r);
            if (p < T(0)) s = -s;
            if (s != T(0)) {
                if (k == m) {
                    if (l != m) a[k * LD + k - 1] = -a[k * LD + k - 1];
                } else
                    a[k * LD + k - 1] = -s * x;
                p += s; x = p / s; y = q / s; z = r / s; q /= p; r /= p;
                for (int j = k; j <= nn; j++) {
                    p = a[k * LD + j] + q * a[(k + 1) * LD + j];
                    if (k != nn - 1) { p += r * a[(k + 2) * LD + j]; a[(k + 2) * LD + j] -= p * z; }
                    a[(k + 1) * LD + j] -= p * y;
                    a[k * LD + j] -= p * x;
                }
                const int mmin = nn < k + 3 ? nn : k + 3;
                for (int i = l; i <= mmin; i++) {
                    p = x * a[i * LD + k] + y * a[i * LD + k + 1];
                    if (k != nn - 1) { p += z * a[i * LD + k + 2]; a[i * LD + k + 2] -= p * r; }
                    a[i * LD + k + 1] -= p * q;
                    a[i * LD + k] -= p;
                }
            }
        }
    }
}

template <typename T, typename IO, int LD>
__global__ void __launch_bounds__(64) vanloan_kernel(int n, int q, int64_t N, VlSrc sA, VlSrc sG, VlSrc sW, VlSrc sdt,
                                                    IO *F, IO *Q, int64_t o_es, int64_t o_fs, uint32_t *status) {
    const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (f >= N) return;
    const IO *pA = (const IO *)sA.p + f * sA.fs, *pG = (const IO *)sG.p + f * sG.fs, *pW = (const IO *)sW.p + f * sW.fs;
    const T dt = (T)((const IO *)sdt.p)[f * sdt.fs];
    T M[LD * LD], H[LD * LD];
    uint32_t st = 0;
    // Nyquist test (c2d.go:16-28)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) H[i * LD + j] = (T)pA[(i * n + j) * sA.es];
    const T lam = vl_last_eig_abs<T, LD>(n, H);
    if (!(T(2) * lam * dt < T(3.14159265358979323846))) st |= KB_ST_NYQUIST;
    // Gamma W Gamma^T dt (c2d.go:31-34): H <- Gamma W (n x q), M12 <- H Gamma^T * dt
    const int N2 = 2 * n;
    for (int i = 0; i < N2; i++)
        for (int j = 0; j < N2; j++) M[i * LD + j] = T(0);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < q; j++) {
            T s = T(0);
            for (int k = 0; k < q; k++) s += (T)pG[(i * q + k) * sG.es] * (T)pW[(k * q + j) * sW.es];
            H[i * LD + j] = s;
        }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            T s = T(0);
            for (int k = 0; k < q; k++) s += H[i * LD + k] * (T)pG[(j * q + k) * sG.es];
            M[i * LD + n + j] = dt * s;
            const T ap = dt * (T)pA[(i * n + j) * sA.es];
            M[i * LD + j] = -ap;                 // :46
            M[(n + j) * LD + n + i] = ap;        // :47  Ap^T
        }
    if (vl_expm<T, LD>(N2, M)) st |= KB_ST_SINGULAR;
    // F = (E22)^T, Q = F * E12 (c2d.go:60-72); AsSymDense: upper triangle mirrored, nil when asymmetric
    bool finite = true;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            T s = T(0);
            for (int k = 0; k < n; k++) s += M[(n + k) * LD + n + i] * M[k * LD + n + j];
            H[i * LD + j] = s;
            finite = finite && (s - s == T(0));
        }
    bool asym = false;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            if (i != j && !sym_close(H[j * LD + i], H[i * LD + j])) asym = true;
            F[(i * n + j) * o_es + f * o_fs] = (IO)M[(n + j) * LD + n + i];
            Q[(i * n + j) * o_es + f * o_fs] = (IO)(i <= j ? H[i * LD + j] : H[j * LD + i]);
        }
    if (asym) st |= KB_ST_ASYMMETRIC;
    if (!finite) st |= KB_ST_NONFINITE;
    status[f] = st;
}

template <typename T, typename IO>
static int vl_launch(int n, int q, int64_t N, const VlSrc &sA, const VlSrc &sG, const VlSrc &sW, const VlSrc &sdt, void *F, void *Q,
                     int64_t o_es, int64_t o_fs, uint32_t *status, hipStream_t stream) {
    const dim3 grid((unsigned)((N + 63) / 64)), block(64);
    int device = 0;
    KB_HIP(hipGetDevice(&device));
    const HeavyScope hs(device, stream, n > 4);   // LD = 12 / 16: 9-16 KB of private arrays per lane, see kb_internal.h
    stream = hs.stream;
    if (n <= 2) hipLaunchKernelGGL((vanloan_kernel<T, IO, 4>), grid, block, 0, stream, n, q, N, sA, sG, sW, sdt, (IO *)F, (IO *)Q, o_es, o_fs, status);
    else if (n <= 4) hipLaunchKernelGGL((vanloan_kernel<T, IO, 8>), grid, block, 0, stream, n, q, N, sA, sG, sW, sdt, (IO *)F, (IO *)Q, o_es, o_fs, status);
    else if (n <= 6) hipLaunchKernelGGL((vanloan_kernel<T, IO, 12>), grid, block, 0, stream, n, q, N, sA, sG, sW, sdt, (IO *)F, (IO *)Q, o_es, o_fs, status);
    else hipLaunchKernelGGL((vanloan_kernel<T, IO, 16>), grid, block, 0, stream, n, q, N, sA, sG, sW, sdt, (IO *)F, (IO *)Q, o_es, o_fs, status);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

static int vl_check(int dtype, int n, int q, int64_t N) {
    if (dtype != KB_F64 && dtype != KB_F32) { set_error("unknown dtype %d", dtype); return KB_ERR_INVALID; }
    if (n < 1 || n > 8) { set_error("kb_van_loan: n = %d outside [1, 8]", n); return KB_ERR_UNSUPPORTED; }
    if (q < 1 || q > 2 * n) { set_error("kb_van_loan: q = %d outside [1, 2n]", q); return KB_ERR_UNSUPPORTED; }
    if (N < 1) { set_error("kb_van_loan: N must be >= 1"); return KB_ERR_INVALID; }
    return KB_OK;
}

}  // namespace kb

using namespace kb;

extern "C" {

int kb_van_loan_dev(int dtype, int n, int q, int64_t N, const void *A, const void *Gamma, const void *W, const void *dt, int64_t ld,
                    void *F, void *Q, uint32_t *status, void *stream) {
    if (!A || !Gamma || !W || !dt || !F || !Q || !status) { set_error("null argument"); return KB_ERR_INVALID; }
    int rc = vl_check(dtype, n, q, N);
    if (rc) return rc;
    if (ld < N) { set_error("ld (%lld) < N (%lld)", (long long)ld, (long long)N); return KB_ERR_INVALID; }
    const VlSrc sA{A, ld, 1}, sG{Gamma, ld, 1}, sW{W, ld, 1}, sdt{dt, 0, 1};
    if (dtype == KB_F64) return vl_launch<double, double>(n, q, N, sA, sG, sW, sdt, F, Q, ld, 1, status, (hipStream_t)stream);
    return vl_launch<float, float>(n, q, N, sA, sG, sW, sdt, F, Q, ld, 1, status, (hipStream_t)stream);
}

int kb_van_loan(int device, int dtype, int n, int q, int64_t N, const double *A, const double *Gamma, const double *W, const double *dt,
                int broadcast, double *F, double *Q, uint32_t *status) {
    if (!A || !Gamma || !W || !dt || !F || !Q) { set_error("null argument"); return KB_ERR_INVALID; }
    int rc = vl_check(dtype, n, q, N);
    if (rc) return rc;
    KB_HIP(hipSetDevice(device));
    const size_t eA = (size_t)n * n, eG = (size_t)n * q, eW = (size_t)q * q;
    const size_t cA = (broadcast & 1) ? 1 : (size_t)N, cG = (broadcast & 2) ? 1 : (size_t)N, cW = (broadcast & 4) ? 1 : (size_t)N,
                 cdt = (broadcast & 8) ? 1 : (size_t)N;
    const size_t in_elems = eA * cA + eG * cG + eW * cW + cdt, out_elems = 2 * eA * (size_t)N;
    double *d = nullptr;
    uint32_t *d_st = nullptr;
    KB_HIP(dev_alloc((void **)&d, (in_elems + out_elems) * sizeof(double)));
    if (dev_alloc((void **)&d_st, (size_t)N * sizeof(uint32_t)) != hipSuccess) { (void)dev_free(d); set_error("out of device memory"); return KB_ERR_HIP; }
    double *dA = d, *dG = dA + eA * cA, *dW = dG + eG * cG, *ddt = dW + eW * cW, *dF = ddt + cdt, *dQ = dF + eA * (size_t)N;
    hipError_t e = hipMemcpy(dA, A, eA * cA * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dG, Gamma, eG * cG * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dW, W, eW * cW * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ddt, dt, cdt * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const VlSrc sA{dA, 1, (broadcast & 1) ? 0 : (int64_t)eA}, sG{dG, 1, (broadcast & 2) ? 0 : (int64_t)eG},
            sW{dW, 1, (broadcast & 4) ? 0 : (int64_t)eW}, sdt{ddt, 0, (broadcast & 8) ? 0 : 1};
        // AoS double in and out; the arithmetic runs in `dtype`
        rc = dtype == KB_F64 ? vl_launch<double, double>(n, q, N, sA, sG, sW, sdt, dF, dQ, 1, (int64_t)eA, d_st, nullptr)
                             : vl_launch<float, double>(n, q, N, sA, sG, sW, sdt, dF, dQ, 1, (int64_t)eA, d_st, nullptr);
        if (!rc) e = hipDeviceSynchronize();
    }
    if (!rc && e == hipSuccess) e = hipMemcpy(F, dF, eA * (size_t)N * sizeof(double), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess) e = hipMemcpy(Q, dQ, eA * (size_t)N * sizeof(double), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && status) e = hipMemcpy(status, d_st, (size_t)N * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)dev_free(d);
    (void)dev_free(d_st);
    if (!rc && e != hipSuccess) rc = hip_fail(e, "kb_van_loan");
    return rc;
}

}  // extern "C"
