// One wave per SIMD (forced with a 39 KB LDS block per 64-thread workgroup): issue rate of v_fma_f32 against
// v_pk_fma_f32 for a wave that has the SIMD to itself.  hipcc --offload-arch=gfx950 -O3 scripts/diag_pkfma.hip -o /tmp/diag_pkfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool PK>
__global__ void __launch_bounds__(64, 1) k(float *out, int iters) {
    __shared__ float pad[39 * 256];
    pad[threadIdx.x] = threadIdx.x;
    float s = pad[threadIdx.x] * 1e-9f + 1.0f;
    if constexpr (PK) {
        v2f acc[16];
        for (int i = 0; i < 16; i++) acc[i] = v2f{s + i, s - i};
        const v2f m = v2f{1.0000001f, 0.9999999f};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = __builtin_elementwise_fma(acc[i], m, v2f{1e-9f, 1e-9f});
        float r = 0;
        for (int i = 0; i < 16; i++) r += acc[i].x + acc[i].y;
        out[blockIdx.x * 64 + threadIdx.x] = r;
    } else {
        float acc[32];
        for (int i = 0; i < 32; i++) acc[i] = s + i;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 32; i++) acc[i] = __builtin_fmaf(acc[i], 1.0000001f, 1e-9f);
        float r = 0;
        for (int i = 0; i < 32; i++) r += acc[i];
        out[blockIdx.x * 64 + threadIdx.x] = r;
    }
}

int main() {
    float *d; hipMalloc(&d, 1024 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int pk = 0; pk < 2; pk++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(k<true>, dim3(1024), dim3(64), 0, 0, d, iters);
            else hipLaunchKernelGGL(k<false>, dim3(1024), dim3(64), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s rep %d: %.3f ms  -> %.2f cycles@2.4GHz per 64-lane FMA (2 for a packed instruction)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", rep, ms, ms * 1e-3 * 2.4e9 / (32.0 * iters));
        }
    }
    return 0;
}
