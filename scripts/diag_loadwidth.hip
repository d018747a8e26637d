// Load-phase model of the fused SRIF kernel: one wave per SIMD (39 KB of LDS per 64-thread workgroup), each wave reads
// 340 floats per lane per tile and stores 90, either as 340 dword loads from an [element][lane] layout or as 85 dwordx4
// loads from an [element/4][lane][4] layout.  hipcc --offload-arch=gfx950 -O3 scripts/diag_loadwidth.hip -o diag_loadwidth
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int W>   // W = floats per load (1 or 4)
__global__ void __launch_bounds__(64, 1) k(const float *__restrict__ in, float *__restrict__ out, int ntiles) {
    __shared__ float pad[39 * 256];
    pad[threadIdx.x] = 0.f;
    const int lane = threadIdx.x;
    const long tile = blockIdx.x;
    if (tile >= ntiles) return;
    const float *p = in + tile * (340L * 64);
    float acc = pad[lane];
    if constexpr (W == 1) {
        float v[340];
#pragma unroll
        for (int e = 0; e < 340; e++) v[e] = __builtin_nontemporal_load(p + e * 64 + lane);
#pragma unroll
        for (int e = 0; e < 340; e++) acc += v[e];
    } else {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v[85];
#pragma unroll
        for (int e = 0; e < 85; e++) v[e] = __builtin_nontemporal_load((const f4 *)p + e * 64 + lane);
#pragma unroll
        for (int e = 0; e < 85; e++) acc += v[e].x + v[e].y + v[e].z + v[e].w;
    }
    float *o = out + tile * (90L * 64);
#pragma unroll
    for (int e = 0; e < 90; e++) o[e * 64 + lane] = acc + e;
}

int main() {
    const int ntiles = 4096;
    float *in, *out;
    hipMalloc(&in, (size_t)ntiles * 340 * 64 * 4);
    hipMalloc(&out, (size_t)ntiles * 90 * 64 * 4);
    hipMemset(in, 0, (size_t)ntiles * 340 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++)
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; i++) {
                if (w == 0) hipLaunchKernelGGL(k<1>, dim3(ntiles), dim3(64), 0, 0, in, out, ntiles);
                else hipLaunchKernelGGL(k<4>, dim3(ntiles), dim3(64), 0, 0, in, out, ntiles);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("%s rep %d: %.1f us per launch, %.2f TB/s (read 357 MB + write 94 MB)\n", w ? "dwordx4" : "dword  ", rep, ms * 1e3, (double)ntiles * 430 * 64 * 4 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
