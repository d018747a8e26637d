// kb_mc.hip -- Monte-Carlo fan-out (montecarlo.go:92-119) of a pure-predictor Vanilla filter
// with AWGN process noise, and its per-step statistics (montecarlo.go:18-59).
//
// The reference runs `samples` runs sequentially on one filter object, Reset() between
// runs (fresh noise seed), storing every Estimate; Mean(k)/StdDev(k) then gather state
// component i over runs.  Here one lane = one run, all `steps` steps inside one launch:
//   x_{k+1} = F x_k [+ G u_k] + L_Q z_k          (vanilla.go:138-146, predictionOnly)
// and per step the wave reduces sum(d) and sum(d^2), d = x - c_k, with c_k the noise-free
// trajectory (same for every run: it removes the mean before squaring, so the unbiased
// variance does not cancel catastrophically when |mean| >> stddev, as in statOD5044).
// Partial sums go to one of REPL replicas by float64 atomics; the host adds the replicas.
// P is not propagated: it is identical for every run and does not influence x; the
// batch is left Reset(), as the reference leaves its filter (montecarlo.go:116).
#include "kb_internal.h"

namespace kb {

constexpr int MC_REPL = 32;


// Sums V values (V a power of two <= 64) over the 64 lanes with V - 1 + log2(64 / V) exchanges instead of 6 V: while
// more than one value is left, a lane keeps one of each pair of values and hands the other to its partner, so every
// exchange halves the number of values per lane.  On return lane l holds the wave total of value l & (V - 1) in v[0].
template <int V>
__device__ __forceinline__ void wave_sum_multi(double (&v)[V], int lane) {
    int off = 1;
#pragma unroll
    for (int nv = V; nv > 1; nv >>= 1, off <<= 1) {
        const bool hi = (lane & off) != 0;
#pragma unroll
        for (int j = 0; j < nv / 2; j++) {
            const double send = hi ? v[2 * j] : v[2 * j + 1];
            const double keep = hi ? v[2 * j + 1] : v[2 * j];
            v[j] = keep + __shfl_xor(send, off, 64);
        }
    }
#pragma unroll
    for (int o = V; o < 64; o <<= 1) v[0] += __shfl_xor(v[0], o, 64);
}

// KEEP: every run's estimate is also written out per step (MonteCarloRun.Estimates[k], montecarlo.go:108-117, as far as it
// differs between runs): State() = x_k and Measurement() = yhat_k = H x_{k-1} + v_k (vanilla.go:155-157; the draw the
// chi-square replay of kb_chisq.hip makes), traj[(t (n + p) + e) ld + run].  The measurement dimension is a run-time loop
// over the model block: this variant serves the reference-sized ensembles (50 x 120, 15 x 1086), not the benchmark.
template <typename T, int NS, int NC, bool KEEP = false>
__global__ void __launch_bounds__(256) mc_kernel(const StepArgs a, const T *__restrict__ controls, int ncontrols,
                                                 double *__restrict__ sums /* [REPL][steps][2][NS] */,
                                                 double *__restrict__ shift /* [steps][NS] */,
                                                 T *__restrict__ traj = nullptr, int64_t traj_ld = 0) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    const T *st = (const T *)a.state + tile * ((int64_t)KB_TILE * a.L.st_elems) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    T x[NS], c[NS], F[NS * NS], LQ[tri(NS)];
    [[maybe_unused]] T G[NC > 0 ? NS * NC : 1];
#pragma unroll
    for (int i = 0; i < NS; i++) { x[i] = ldt(st, a.L.st_vec + i); c[i] = x[i]; }
#pragma unroll
    for (int e = 0; e < NS * NS; e++) F[e] = ldt(mo, a.L.mo_F + e);
#pragma unroll
    for (int e = 0; e < tri(NS); e++) LQ[e] = ldt(mo, a.L.mo_LQ + e);
    if constexpr (NC > 0) {
#pragma unroll
        for (int e = 0; e < NS * NC; e++) G[e] = ldt(mo, a.L.mo_G + e);
    }
    double *my = sums + (size_t)(tile % MC_REPL) * a.nsteps * 2 * NS;
    const uint64_t gfi = (uint64_t)(a.first_filter + fi);
    for (int t = 0; t < a.nsteps; t++) {
        if constexpr (KEEP) {
            if (active) {
                T *row = traj + ((int64_t)t * (NS + a.p) + NS) * traj_ld + fi;
                for (int r = 0; r < a.p; r++) {
                    T s = T(0);
#pragma unroll
                    for (int l = 0; l < NS; l++) s += ldt(mo, a.L.mo_H + r * NS + l) * x[l];   // yhat = H x_prev ...
                    T v = T(0);
                    for (int k = 0; k <= r; k++)                                                // ... + v_k, v = L_R z (Noise.Measurement(k))
                        v += ldt(mo, a.L.mo_LR + symi(k, r)) * (T)normal_at(a.seed, gfi, (uint32_t)(a.step0 + t), (uint32_t)(a.epoch * 4 + 1), k);
                    row[(int64_t)r * traj_ld] = s + v;
                }
            }
        }
        T xn[NS], cn[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0), sc = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) { s += F[i * NS + l] * x[l]; sc += F[i * NS + l] * c[l]; }
            xn[i] = s; cn[i] = sc;
        }
        if constexpr (NC > 0) {
            const T *u = controls + (ncontrols == 1 ? 0 : (int64_t)t * NC);
#pragma unroll
            for (int i = 0; i < NS; i++) {
                T s = T(0);
#pragma unroll
                for (int k = 0; k < NC; k++) s += G[i * NC + k] * (ncontrols == 1 ? T(0) : u[k]);
                xn[i] = xn[i] + s; cn[i] = cn[i] + s;
            }
        }
        // Noise.Process(k): w = L_Q z  (noise.go:133-136; distmv.Normal.Rand = mu + L z)
        T z[NS];
#pragma unroll
        for (int k = 0; k < NS; k += 2) {
            uint32_t r[4];
            Philox::gen(a.seed, gfi, (uint32_t)(a.step0 + t), ((uint32_t)(a.epoch * 4 + 0) << 8) | (uint32_t)(k >> 1), r);
            double z0, z1;
            box_muller(r, z0, z1);
            z[k] = (T)z0;
            if (k + 1 < NS) z[k + 1] = (T)z1;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int k = 0; k <= i; k++) s += LQ[symi(k, i)] * z[k];
            x[i] = xn[i] + s;
            c[i] = cn[i];
        }
        if constexpr (KEEP) {
            if (active) {
#pragma unroll
                for (int i = 0; i < NS; i++) traj[((int64_t)t * (NS + a.p) + i) * traj_ld + fi] = x[i];
            }
        }
        constexpr int V = NS <= 2 ? 4 : (NS <= 4 ? 8 : 16);   // 2 NS values, padded to a power of two
        double acc[V];
#pragma unroll
        for (int i = 0; i < V / 2; i++) {
            const double dlt = (i < NS && active) ? (double)x[i < NS ? i : 0] - (double)c[i < NS ? i : 0] : 0.0;
            acc[2 * i] = dlt;
            acc[2 * i + 1] = dlt * dlt;
        }
        wave_sum_multi<V>(acc, lane);
        if (lane < 2 * NS) atomicAdd(my + ((size_t)t * 2 + (lane & 1)) * NS + (lane >> 1), acc[0]);   // one instruction, 2 NS lanes
        if (tile == 0 && lane == 0) {
#pragma unroll
            for (int i = 0; i < NS; i++) shift[(size_t)t * NS + i] = (double)c[i];
        }
    }
}

template <typename T, int NS>
static bool mc_try(const Batch &b, const StepArgs &a, const void *d_controls, int ncontrols, double *d_sums, double *d_shift, void *traj, int64_t traj_ld) {
    if (a.n != NS) return false;
    const int nc = a.need_ctrl ? a.m : 0;
    const dim3 grid = tile_grid(a.ntiles), block(256);
#define KB_MC(NC_) do { \
        if (traj) hipLaunchKernelGGL((mc_kernel<T, NS, NC_, true>), grid, block, 0, b.stream, a, (const T *)d_controls, ncontrols, d_sums, d_shift, (T *)traj, traj_ld); \
        else hipLaunchKernelGGL((mc_kernel<T, NS, NC_, false>), grid, block, 0, b.stream, a, (const T *)d_controls, ncontrols, d_sums, d_shift, (T *)nullptr, (int64_t)0); \
        return true; } while (0)
    switch (nc) {
    case 0: KB_MC(0);
    case 1: KB_MC(1);
    case 2: KB_MC(2);
    }
#undef KB_MC
    return false;
}

// Any other state dimension up to 16 (montecarlo.go:92-119 is shape-generic).  The ensemble is ONE filter fanned out (kb_replicate), so
// the model is the same for every run: F, G, chol(Q), H, chol(R) are read once per workgroup into LDS (run 0's copy) and every lane
// reads them from there as broadcast operands -- per lane only x and the noise-free trajectory c live in registers (2 n doubles;
// with the model per lane, as in mc_kernel, 12 states would need 250).  Same sums in the same order as mc_kernel; zero padding to NS.
template <typename T, int NS>
__global__ void __launch_bounds__(256) mc_gen_kernel(const StepArgs a, const T *__restrict__ controls, int ncontrols,
                                                     double *__restrict__ sums /* [REPL][steps][2][n] */, double *__restrict__ shift /* [steps][n] */,
                                                     T *__restrict__ traj, int64_t traj_ld) {
    constexpr int TQ = tri(NS), PM = 8, TP = tri(PM);   // (p <= 8: the engine's envelope for n <= 16)
    __shared__ T sF[NS * NS], sLQ[TQ], sG[NS * 2], sH[PM * NS], sLR[TP];
    const int n = a.n, p = a.p, nc = a.need_ctrl ? a.m : 0;
    {   // run 0's model block: element e of a field at mo0[(field + e) * KB_TILE]
        const T *mo0 = (const T *)a.model;
        for (int e = threadIdx.x; e < NS * NS; e += blockDim.x) { const int i = e / NS, l = e % NS; sF[e] = (i < n && l < n) ? ldt(mo0, a.L.mo_F + i * n + l) : T(0); }
        for (int e = threadIdx.x; e < TQ; e += blockDim.x) sLQ[e] = e < tri(n) ? ldt(mo0, a.L.mo_LQ + e) : T(0);   // (packed by rows of the lower triangle: the index does not depend on n)
        for (int e = threadIdx.x; e < NS * 2; e += blockDim.x) { const int i = e / 2, c = e % 2; sG[e] = (i < n && c < nc) ? ldt(mo0, a.L.mo_G + i * nc + c) : T(0); }
        if (traj) {
            for (int e = threadIdx.x; e < PM * NS; e += blockDim.x) { const int r = e / NS, l = e % NS; sH[e] = (r < p && l < n) ? ldt(mo0, a.L.mo_H + r * n + l) : T(0); }
            for (int e = threadIdx.x; e < TP; e += blockDim.x) sLR[e] = e < tri(p) ? ldt(mo0, a.L.mo_LR + e) : T(0);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    const T *st = (const T *)a.state + tile * ((int64_t)KB_TILE * a.L.st_elems) + lane;
    T x[NS], c[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) { x[i] = i < n ? ldt(st, a.L.st_vec + i) : T(0); c[i] = x[i]; }
    double *my = sums + (size_t)(tile % MC_REPL) * a.nsteps * 2 * n;
    const uint64_t gfi = (uint64_t)(a.first_filter + fi);
    for (int t = 0; t < a.nsteps; t++) {
        asm volatile("" ::: "memory");   // the model is re-read from LDS every step, not hoisted into (hundreds of) registers
        if (traj && active) {   // Measurement() = H x_prev + L_R z (vanilla.go:155-157), Noise.Measurement(k)
            for (int r = 0; r < p; r++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += sH[r * NS + l] * x[l];
                T v = T(0);
                for (int k = 0; k <= r; k++) v += sLR[symi(k, r)] * (T)normal_at(a.seed, gfi, (uint32_t)(a.step0 + t), (uint32_t)(a.epoch * 4 + 1), k);
                traj[((int64_t)t * (n + p) + n + r) * traj_ld + fi] = s + v;
            }
        }
        T xn[NS], cn[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0), sc = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) { const T f = sF[i * NS + l]; s += f * x[l]; sc += f * c[l]; }
            xn[i] = s; cn[i] = sc;
        }
        if (nc > 0) {
            const T *u = controls + (ncontrols == 1 ? 0 : (int64_t)t * nc);
            const T u0 = ncontrols == 1 ? T(0) : u[0], u1 = (ncontrols == 1 || nc < 2) ? T(0) : u[1];
#pragma unroll
            for (int i = 0; i < NS; i++) {
                T s = T(0);
                s += sG[i * 2 + 0] * u0;
                s += sG[i * 2 + 1] * u1;
                xn[i] = xn[i] + s; cn[i] = cn[i] + s;
            }
        }
        T z[NS];   // Noise.Process(k): w = L_Q z (noise.go:133-136)
#pragma unroll
        for (int k = 0; k < NS; k += 2) {
            z[k] = T(0);
            if (k + 1 < NS) z[k + 1] = T(0);
            if (k < n) {   // (wave-uniform)
                uint32_t r[4];
                Philox::gen(a.seed, gfi, (uint32_t)(a.step0 + t), ((uint32_t)(a.epoch * 4 + 0) << 8) | (uint32_t)(k >> 1), r);
                double z0, z1;
                box_muller(r, z0, z1);
                z[k] = (T)z0;
                if (k + 1 < NS) z[k + 1] = (T)z1;
            }
        }
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int k = 0; k <= i; k++) s += sLQ[symi(k, i)] * z[k];
            x[i] = xn[i] + s;
            c[i] = cn[i];
        }
        if (traj && active) {
#pragma unroll
            for (int i = 0; i < NS; i++)
                if (i < n) traj[((int64_t)t * (n + p) + i) * traj_ld + fi] = x[i];
        }
        constexpr int V = NS <= 8 ? 16 : 32;   // 2 NS values, padded to a power of two
        double acc[V];
#pragma unroll
        for (int i = 0; i < V / 2; i++) {
            const double dlt = (i < NS && active) ? (double)x[i < NS ? i : 0] - (double)c[i < NS ? i : 0] : 0.0;
            acc[2 * i] = dlt;
            acc[2 * i + 1] = dlt * dlt;
        }
        wave_sum_multi<V>(acc, lane);
        if (lane < 2 * n) atomicAdd(my + ((size_t)t * 2 + (lane & 1)) * n + (lane >> 1), acc[0]);
        if (tile == 0 && lane == 0) {
#pragma unroll
            for (int i = 0; i < NS; i++)
                if (i < n) shift[(size_t)t * n + i] = (double)c[i];
        }
    }
}

template <typename T, int NS>
static bool mc_gen_try(const Batch &b, const StepArgs &a, const void *d_controls, int ncontrols, double *d_sums, double *d_shift, void *traj, int64_t traj_ld) {
    if (a.n > NS || a.p > 8 || (a.need_ctrl ? a.m : 0) > 2) return false;
    hipLaunchKernelGGL((mc_gen_kernel<T, NS>), tile_grid(a.ntiles), dim3(256), 0, b.stream, a, (const T *)d_controls, ncontrols, d_sums, d_shift, (T *)traj, traj_ld);
    return true;
}

template <typename T>
static int launch_mc_t(const Batch &b, const StepArgs &a, const void *d_controls, int ncontrols, double *d_sums, double *d_shift, void *traj, int64_t traj_ld) {
    const bool ok = mc_try<T, 2>(b, a, d_controls, ncontrols, d_sums, d_shift, traj, traj_ld) || mc_try<T, 3>(b, a, d_controls, ncontrols, d_sums, d_shift, traj, traj_ld) ||
                    mc_try<T, 4>(b, a, d_controls, ncontrols, d_sums, d_shift, traj, traj_ld) || mc_try<T, 6>(b, a, d_controls, ncontrols, d_sums, d_shift, traj, traj_ld) ||
                    mc_gen_try<T, 8>(b, a, d_controls, ncontrols, d_sums, d_shift, traj, traj_ld) || mc_gen_try<T, 12>(b, a, d_controls, ncontrols, d_sums, d_shift, traj, traj_ld) ||
                    mc_gen_try<T, 16>(b, a, d_controls, ncontrols, d_sums, d_shift, traj, traj_ld);
    if (!ok) {
        set_error("kb_mc_run: no Monte-Carlo kernel for n=%d, p=%d, m=%d (built: n <= 16, p <= 8, m <= 2)", a.n, a.p, a.need_ctrl ? a.m : 0);
        return KB_ERR_UNSUPPORTED;
    }
    KB_HIP(hipGetLastError());
    return KB_OK;
}

// d_sums: [MC_REPL][steps][2][n] followed by shift [steps][n]
int launch_mc(const Batch &b, const StepArgs &a, const void *d_controls, int ncontrols, double *d_sums, void *traj, int64_t traj_ld) {
    double *d_shift = d_sums + (size_t)MC_REPL * a.nsteps * 2 * a.n;
    if (b.dtype == KB_F64) return launch_mc_t<double>(b, a, d_controls, ncontrols, d_sums, d_shift, traj, traj_ld);
    return launch_mc_t<float>(b, a, d_controls, ncontrols, d_sums, d_shift, traj, traj_ld);
}

int mc_repl() { return MC_REPL; }

// out[i] = sum over the replicas r of src[r * per + i] (replica order, so the result does not depend on who asks): the partial sums
// of a launch folded on the device, where a multi-device driver can all-reduce them in place (kb_sharded.hip)
__global__ void fold_replicas_kernel(const double *__restrict__ src, int repl, int64_t per, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    double s = 0.0;
    for (int r = 0; r < repl; r++) s += src[(int64_t)r * per + i];
    out[i] = s;
}

int launch_fold(hipStream_t stream, const double *src, int repl, int64_t per, double *out) {
    hipLaunchKernelGGL(fold_replicas_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, stream, src, repl, per, out);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
