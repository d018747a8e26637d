// Issue rate of v_fma_f32 against v_pk_fma_f32 with 1, 2 and 4 waves per SIMD (an LDS block per 64-thread workgroup sets the
// occupancy: 160 KB per CU / (4 SIMDs x waves)).  hipcc --offload-arch=gfx950 -O3 scripts/diag_pkfma.hip -o /tmp/diag_pkfma
// Prints SIMD cycles (at the nominal 2.4 GHz) per 64-lane FMA: a packed instruction counts as two.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool PK, int LDSKB>
__global__ void __launch_bounds__(64) k(float *out, int iters) {
    __shared__ float pad[LDSKB * 256];
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

template <bool PK, int LDSKB>
static void run(float *d, int waves, const char *name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<PK, LDSKB>), dim3(1024 * waves), dim3(64), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    // one SIMD runs `waves` waves of 32 FMAs x iters each
    printf("%s, %d wave(s) per SIMD: %.3f ms -> %.2f SIMD cycles@2.4GHz per 64-lane FMA\n", name, waves, best, best * 1e-3 * 2.4e9 / (32.0 * iters * waves));
}

int main() {
    float *d; hipMalloc(&d, 4096 * 64 * 4);
    run<false, 39>(d, 1, "v_fma_f32   "); run<true, 39>(d, 1, "v_pk_fma_f32");
    run<false, 19>(d, 2, "v_fma_f32   "); run<true, 19>(d, 2, "v_pk_fma_f32");
    run<false, 9>(d, 4, "v_fma_f32   ");  run<true, 9>(d, 4, "v_pk_fma_f32");
    return 0;
}
