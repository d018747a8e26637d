// diag_stream.hip -- diagnostic (not product): what does the memory system deliver for the
// Vanilla kernel's access pattern with the arithmetic removed?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/diag scripts/diag_stream.hip && /tmp/diag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ST = 27, MO = 81, YE = 3;  // elements per filter: state block, model block (F,H,Q,R packed), y

// pattern A: one filter per lane, 8 B per lane per load (512 B rows), all loads then reduce then stores
template <int WAVES_PER_BLOCK>
__global__ void __launch_bounds__(WAVES_PER_BLOCK * 64) pat_rows8(double *st, const double *mo, const double *y, int64_t ntiles, int64_t N) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    double *s = st + tile * 64 * ST + lane;
    const double *m = mo + tile * 64 * MO + lane;
    double v[ST + MO + YE];
#pragma unroll
    for (int e = 0; e < ST; e++) v[e] = s[e * 64];
#pragma unroll
    for (int e = 0; e < MO; e++) v[ST + e] = m[e * 64];
#pragma unroll
    for (int e = 0; e < YE; e++) v[ST + MO + e] = y[(int64_t)e * N + tile * 64 + lane];
    double acc = 0;
#pragma unroll
    for (int e = ST; e < ST + MO + YE; e++) acc += v[e];
#pragma unroll
    for (int e = 0; e < ST; e++) s[e * 64] = v[e] + acc;
}

// pattern A with cache-policy variants: NTL = non-temporal model/y loads, NTS = non-temporal state stores
template <bool NTL, bool NTS, bool NTSL>
__global__ void __launch_bounds__(256) pat_rows8_nt(double *st, const double *mo, const double *y, int64_t ntiles, int64_t N) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    double *s = st + tile * 64 * ST + lane;
    const double *m = mo + tile * 64 * MO + lane;
    double v[ST + MO + YE];
#pragma unroll
    for (int e = 0; e < ST; e++) v[e] = NTSL ? __builtin_nontemporal_load(s + e * 64) : s[e * 64];
#pragma unroll
    for (int e = 0; e < MO; e++) v[ST + e] = NTL ? __builtin_nontemporal_load(m + e * 64) : m[e * 64];
#pragma unroll
    for (int e = 0; e < YE; e++) v[ST + MO + e] = NTL ? __builtin_nontemporal_load(y + (int64_t)e * N + tile * 64 + lane) : y[(int64_t)e * N + tile * 64 + lane];
    double acc = 0;
#pragma unroll
    for (int e = ST; e < ST + MO + YE; e++) acc += v[e];
#pragma unroll
    for (int e = 0; e < ST; e++) { if (NTS) __builtin_nontemporal_store(v[e] + acc, s + e * 64); else s[e * 64] = v[e] + acc; }
}

// pattern B: same bytes, 16 B per lane per load (two rows per wave-instruction)
template <int WAVES_PER_BLOCK>
__global__ void __launch_bounds__(WAVES_PER_BLOCK * 64) pat_rows16(double *st, const double *mo, const double *y, int64_t ntiles, int64_t N) {
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    double2 *s = (double2 *)(st + tile * 64 * 28) + lane;          // 28 rows (27 padded to even)
    const double2 *m = (const double2 *)(mo + tile * 64 * 82) + lane;  // 82 rows
    double2 v[14 + 41];
#pragma unroll
    for (int e = 0; e < 14; e++) v[e] = s[e * 64];
#pragma unroll
    for (int e = 0; e < 41; e++) v[14 + e] = m[e * 64];
    double acc = y[tile * 64 + lane] + y[N + tile * 64 + lane] + y[2 * N + tile * 64 + lane];
#pragma unroll
    for (int e = 14; e < 55; e++) acc += v[e].x + v[e].y;
#pragma unroll
    for (int e = 0; e < 14; e++) { double2 o; o.x = v[e].x + acc; o.y = v[e].y + acc; s[e * 64] = o; }
}

// pattern C: plain streaming copy, 16 B per lane
__global__ void copy16(const double2 *__restrict__ a, double2 *__restrict__ b, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}
// pattern D: read-only streaming (sum), 16 B per lane
__global__ void read16(const double2 *__restrict__ a, double *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (; i < n; i += stride) { double2 v = a[i]; acc += v.x + v.y; }
    if (acc == 1.2345e300) out[0] = acc;
}

int main() {
    const int64_t N = 1 << 20, ntiles = N / 64;
    double *st, *mo, *y, *big;
    CK(hipMalloc(&st, N * 28 * 8)); CK(hipMalloc(&mo, N * 82 * 8)); CK(hipMalloc(&y, N * 3 * 8));
    const int64_t nb = (int64_t)1 << 27;  // 2^27 double2 = 2 GiB
    CK(hipMalloc(&big, nb * 16 * 2));
    CK(hipMemset(st, 0, N * 28 * 8)); CK(hipMemset(mo, 0, N * 82 * 8)); CK(hipMemset(y, 0, N * 3 * 8)); CK(hipMemset(big, 0, nb * 32));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double bytes, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / (ms * 1e-3) / 1e9);
    };
    const double bytesA = (double)N * 8 * (ST + MO + YE + ST);
    timeit("rows8  4 waves/block (kernel's pattern)", bytesA, [&] { hipLaunchKernelGGL(pat_rows8<4>, dim3(ntiles / 4), dim3(256), 0, 0, st, mo, y, ntiles, N); });
    timeit("rows8  1 wave/block", bytesA, [&] { hipLaunchKernelGGL(pat_rows8<1>, dim3(ntiles), dim3(64), 0, 0, st, mo, y, ntiles, N); });
    timeit("rows8 nt model loads", bytesA, [&] { hipLaunchKernelGGL((pat_rows8_nt<true, false, false>), dim3(ntiles / 4), dim3(256), 0, 0, st, mo, y, ntiles, N); });
    timeit("rows8 nt state stores", bytesA, [&] { hipLaunchKernelGGL((pat_rows8_nt<false, true, false>), dim3(ntiles / 4), dim3(256), 0, 0, st, mo, y, ntiles, N); });
    timeit("rows8 nt model loads + nt state stores", bytesA, [&] { hipLaunchKernelGGL((pat_rows8_nt<true, true, false>), dim3(ntiles / 4), dim3(256), 0, 0, st, mo, y, ntiles, N); });
    timeit("rows8 nt everything", bytesA, [&] { hipLaunchKernelGGL((pat_rows8_nt<true, true, true>), dim3(ntiles / 4), dim3(256), 0, 0, st, mo, y, ntiles, N); });
    const double bytesB = (double)N * 8 * (28 + 82 + 3 + 28);
    timeit("rows16 4 waves/block", bytesB, [&] { hipLaunchKernelGGL(pat_rows16<4>, dim3(ntiles / 4), dim3(256), 0, 0, st, mo, y, ntiles, N); });
    timeit("rows16 1 wave/block", bytesB, [&] { hipLaunchKernelGGL(pat_rows16<1>, dim3(ntiles), dim3(64), 0, 0, st, mo, y, ntiles, N); });
    timeit("copy16 2 GiB -> 2 GiB (R+W bytes)", (double)nb * 32, [&] { hipLaunchKernelGGL(copy16, dim3(256 * 8), dim3(256), 0, 0, (const double2 *)big, (double2 *)big + nb, nb); });
    timeit("read16 2 GiB", (double)nb * 16, [&] { hipLaunchKernelGGL(read16, dim3(256 * 8), dim3(256), 0, 0, (const double2 *)big, st, nb); });
    timeit("copy16 80/20 mix: read 1.6 GiB + copy 0.4 GiB", (double)nb * 16 * 0.8 + (double)nb * 32 * 0.2, [&] {
        hipLaunchKernelGGL(read16, dim3(256 * 8), dim3(256), 0, 0, (const double2 *)big, st, (int64_t)(nb * 0.8));
        hipLaunchKernelGGL(copy16, dim3(256 * 8), dim3(256), 0, 0, (const double2 *)big, (double2 *)big + nb, (int64_t)(nb * 0.2)); });
    return 0;
}
