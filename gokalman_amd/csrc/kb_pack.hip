// kb_pack.hip -- host-layout (per-filter row-major) <-> AoSoA-64 block transposition.
// Set-up / read-back path only; the step kernels never touch the host layout.
#include "kb_internal.h"

namespace kb {

struct MapArg { int16_t m[KB_MAX_DIM * KB_MAX_DIM]; };

// src: AoS [count][src_elems] (always float64, converted to T here), one thread per filter.  dst elem = map[e] (or skip if < 0).
template <typename T>
__global__ void pack_kernel(const double *__restrict__ src, int src_elems, int broadcast, int64_t N,
                            T *__restrict__ dst, int dst_elems, MapArg map) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double *s = src + (broadcast ? 0 : i * src_elems);
    T *d = dst + (i / KB_TILE) * ((int64_t)KB_TILE * dst_elems) + (i % KB_TILE);
    for (int e = 0; e < src_elems; e++) {
        const int de = map.m[e];
        if (de >= 0) d[(int64_t)de * KB_TILE] = (T)s[e];
    }
}

// dst: AoS double [count][dst_elems]; src elem = map[e] (or 0.0 if < 0)
template <typename T>
__global__ void unpack_kernel(const T *__restrict__ src, int src_elems, int64_t first, int64_t count,
                              double *__restrict__ dst, int dst_elems, MapArg map) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int64_t i = first + k;
    const T *s = src + (i / KB_TILE) * ((int64_t)KB_TILE * src_elems) + (i % KB_TILE);
    double *d = dst + k * dst_elems;
    for (int e = 0; e < dst_elems; e++) {
        const int se = map.m[e];
        d[e] = se >= 0 ? (double)s[(int64_t)se * KB_TILE] : 0.0;
    }
}

template <typename T>
__global__ void unpack_planar_kernel(const T *__restrict__ src, int src_elems, int64_t N,
                                     T *__restrict__ dst, int64_t ld, int dst_elems, MapArg map) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const T *s = src + (i / KB_TILE) * ((int64_t)KB_TILE * src_elems) + (i % KB_TILE);
    for (int e = 0; e < dst_elems; e++) {
        const int se = map.m[e];
        dst[(int64_t)e * ld + i] = se >= 0 ? s[(int64_t)se * KB_TILE] : T(0);
    }
}

// Estimate snapshot (kb_get_estimate): up to six members and the status words of filters [first, first + count) in ONE launch.
// dst members are AoS double [count][out_elems] at byte offsets of `area`; status is read (and cleared) atomically.
// One workgroup (one wave) per filter, the lanes share the elements: for the one-filter batches of the drop-in path the
// destination is pinned HOST memory written across PCIe, where 64 lanes storing neighbouring doubles make a few large
// transactions instead of a hundred 8-byte ones.
template <typename T>
__global__ void __launch_bounds__(64) snapshot_kernel(SnapArgs sa, int64_t first, int64_t count, char *area, uint32_t *status, int64_t status_off, int clear,
                                                      uint32_t *done, uint32_t seq) {
  for (int64_t k = blockIdx.x; k < count; k += gridDim.x) {
    const int64_t i = first + k;
    for (int m = 0; m < sa.nmembers; m++) {
        const T *s = (const T *)sa.block[m] + (i / KB_TILE) * ((int64_t)KB_TILE * sa.block_elems[m]) + (i % KB_TILE);
        double *d = (double *)(area + sa.off[m]) + k * sa.out_elems[m];
        for (int e = threadIdx.x; e < sa.out_elems[m]; e += 64) {
            const int se = sa.map[m][e];
            d[e] = se >= 0 ? (double)s[(int64_t)se * KB_TILE] : 0.0;
        }
    }
    if (status && threadIdx.x == 0) {
        const uint32_t v = clear ? atomicExch(status + i, 0u) : status[i];
        ((uint32_t *)(area + status_off))[k] = v;
    }
  }
    if (done) {   // one block (launch_snapshot): everything above is in pinned host memory before the host sees the new sequence number
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int launch_snapshot(const Batch &b, const SnapArgs &sa, int64_t first, int64_t count, void *area, uint32_t *status, int64_t status_off, int clear,
                    uint32_t *done, uint32_t seq) {
    if (count <= 0) return KB_OK;
    const dim3 grid(done ? 1u : (unsigned)count);
    if (b.dtype == KB_F64)
        hipLaunchKernelGGL(snapshot_kernel<double>, grid, dim3(64), 0, b.stream, sa, first, count, (char *)area, status, status_off, clear, done, seq);
    else
        hipLaunchKernelGGL(snapshot_kernel<float>, grid, dim3(64), 0, b.stream, sa, first, count, (char *)area, status, status_off, clear, done, seq);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

// kb_replicate: every filter of dst becomes a copy of filter `sf` of src (same block layout)
template <typename T>
__global__ void replicate_kernel(const T *__restrict__ src, int elems, int64_t sf, T *__restrict__ dst, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const T *s = src + (sf / KB_TILE) * ((int64_t)KB_TILE * elems) + (sf % KB_TILE);
    T *d = dst + (i / KB_TILE) * ((int64_t)KB_TILE * elems) + (i % KB_TILE);
    for (int e = 0; e < elems; e++) d[(int64_t)e * KB_TILE] = s[(int64_t)e * KB_TILE];
}

int launch_replicate(const Batch &dst, const void *src_block, int elems, int64_t src_filter, void *dst_block) {
    const unsigned blocks = (unsigned)((dst.N + 255) / 256);
    if (dst.dtype == KB_F64)
        hipLaunchKernelGGL(replicate_kernel<double>, dim3(blocks), dim3(256), 0, dst.stream, (const double *)src_block, elems, src_filter, (double *)dst_block, dst.N);
    else
        hipLaunchKernelGGL(replicate_kernel<float>, dim3(blocks), dim3(256), 0, dst.stream, (const float *)src_block, elems, src_filter, (float *)dst_block, dst.N);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

// kb_mc_get_runs: trajectory buffer [(t * (n + p) + e) * ld + run] -> host layout [run][step][n] (states) and [run][step][p]
// (measurements), float64.  One thread per (run, step), runs fastest: the reads are coalesced.
template <typename T>
__global__ void traj_unpack_kernel(const T *__restrict__ traj, int64_t ld, int n, int p, int steps, int64_t first, int64_t count,
                                   double *__restrict__ states, double *__restrict__ meas) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * steps) return;
    const int64_t k = idx % count;
    const int t = (int)(idx / count);
    const T *s = traj + ((int64_t)t * (n + p)) * ld + first + k;
    if (states)
        for (int e = 0; e < n; e++) states[(k * steps + t) * n + e] = (double)s[(int64_t)e * ld];
    if (meas)
        for (int e = 0; e < p; e++) meas[(k * steps + t) * p + e] = (double)s[(int64_t)(n + e) * ld];
}

int launch_traj_unpack(const Batch &b, int64_t first, int64_t count, double *d_states, double *d_meas) {
    const int64_t total = count * b.mc_steps;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (b.dtype == KB_F64)
        hipLaunchKernelGGL(traj_unpack_kernel<double>, dim3(blocks), dim3(256), 0, b.stream, (const double *)b.d_traj, b.mc_ld, b.n, b.mc_p, b.mc_steps, first, count, d_states, d_meas);
    else
        hipLaunchKernelGGL(traj_unpack_kernel<float>, dim3(blocks), dim3(256), 0, b.stream, (const float *)b.d_traj, b.mc_ld, b.n, b.mc_p, b.mc_steps, first, count, d_states, d_meas);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

static MapArg make_map(const int16_t *map, int n) {
    MapArg m;
    for (int i = 0; i < KB_MAX_DIM * KB_MAX_DIM; i++) m.m[i] = i < n ? map[i] : (int16_t)-1;
    return m;
}

int launch_pack(const Batch &b, const void *src_aos, int src_elems, int64_t count, bool broadcast,
                void *dst_block, int dst_elems, const int16_t *map) {
    (void)count;
    const MapArg m = make_map(map, src_elems);
    const int threads = 256;
    const unsigned blocks = (unsigned)((b.N + threads - 1) / threads);
    if (b.dtype == KB_F64)
        hipLaunchKernelGGL(pack_kernel<double>, dim3(blocks), dim3(threads), 0, b.stream, (const double *)src_aos,
                           src_elems, broadcast ? 1 : 0, b.N, (double *)dst_block, dst_elems, m);
    else
        hipLaunchKernelGGL(pack_kernel<float>, dim3(blocks), dim3(threads), 0, b.stream, (const double *)src_aos,
                           src_elems, broadcast ? 1 : 0, b.N, (float *)dst_block, dst_elems, m);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

int launch_unpack(const Batch &b, const void *src_block, int src_elems, const int16_t *map, int dst_elems,
                  double *dst_aos, int64_t first, int64_t count) {
    const MapArg m = make_map(map, dst_elems);
    const int threads = 256;
    const unsigned blocks = (unsigned)((count + threads - 1) / threads);
    if (count <= 0) return KB_OK;
    if (b.dtype == KB_F64)
        hipLaunchKernelGGL(unpack_kernel<double>, dim3(blocks), dim3(threads), 0, b.stream, (const double *)src_block,
                           src_elems, first, count, dst_aos, dst_elems, m);
    else
        hipLaunchKernelGGL(unpack_kernel<float>, dim3(blocks), dim3(threads), 0, b.stream, (const float *)src_block,
                           src_elems, first, count, dst_aos, dst_elems, m);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

int launch_unpack_planar(const Batch &b, const void *src_block, int src_elems, const int16_t *map, int dst_elems,
                         void *dst, int64_t ld) {
    const MapArg m = make_map(map, dst_elems);
    const int threads = 256;
    const unsigned blocks = (unsigned)((b.N + threads - 1) / threads);
    if (b.dtype == KB_F64)
        hipLaunchKernelGGL(unpack_planar_kernel<double>, dim3(blocks), dim3(threads), 0, b.stream,
                           (const double *)src_block, src_elems, b.N, (double *)dst, ld, dst_elems, m);
    else
        hipLaunchKernelGGL(unpack_planar_kernel<float>, dim3(blocks), dim3(threads), 0, b.stream,
                           (const float *)src_block, src_elems, b.N, (float *)dst, ld, dst_elems, m);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
