// Access-pattern model for a lane-pair SRIF kernel (two lanes per filter, 32 filters per wave): 340 floats read and 90
// written per filter, 262144 filters, with four lane mappings / layouts:
//   0  one filter per lane, [element][64] tiles (today's kernels; contiguous 256 B per wave-load)
//   1  lane = 2 f + l, [element][64] tiles: lane l of a pair reads elements of parity l -> two interleaved 128-B segments per wave-load
//   2  lane = 2 f + l, pair-interleaved tiles [element / 2][32 filters][2] (contiguous 256 B per wave-load)
//   3  lane = 32 l + f, [element][64] tiles: each half-wave reads one contiguous 128-B segment
// hipcc --offload-arch=gfx950 -O3 scripts/diag_lanepair.hip -o diag_lanepair
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int E = 340, W = 90;

template <int MODE>
__global__ void __launch_bounds__(256) k(const float *__restrict__ in, float *__restrict__ out, long nfilters) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc0 = 0.f, acc1 = 0.f;
    if constexpr (MODE == 0) {
        const long tile = wave;
        if (tile * 64 >= nfilters) return;
        const float *p = in + tile * (long)(E * 64) + lane;
#pragma unroll 20
        for (int e = 0; e < E; e += 2) { acc0 += __builtin_nontemporal_load(p + e * 64); acc1 += __builtin_nontemporal_load(p + (e + 1) * 64); }
        float *o = out + tile * (long)(W * 64) + lane;
#pragma unroll 10
        for (int e = 0; e < W; e++) o[e * 64] = acc0 + acc1 * e;
    } else {
        // a wave owns 32 filters = half a tile
        const long tile = wave >> 1;
        const int half = (int)(wave & 1);
        if (tile * 64 >= nfilters) return;
        int f, l;
        if (MODE == 3) { f = lane & 31; l = lane >> 5; } else { f = lane >> 1; l = lane & 1; }
        if constexpr (MODE == 2) {
            const float *p = in + tile * (long)(E * 64) + half * (E * 32) + lane;   // [q][32][2] per half tile
#pragma unroll 20
            for (int q = 0; q < E / 2; q += 2) { acc0 += __builtin_nontemporal_load(p + q * 64); acc1 += __builtin_nontemporal_load(p + (q + 1) * 64); }
            float *o = out + tile * (long)(W * 64) + half * (W * 32) + lane;
#pragma unroll 9
            for (int q = 0; q < W / 2; q++) o[q * 64] = acc0 + acc1 * q;
        } else {
            const float *p = in + tile * (long)(E * 64) + half * 32 + f + l * 64;
#pragma unroll 20
            for (int q = 0; q < E / 2; q += 2) { acc0 += __builtin_nontemporal_load(p + (2 * q) * 64); acc1 += __builtin_nontemporal_load(p + (2 * q + 2) * 64); }
            float *o = out + tile * (long)(W * 64) + half * 32 + f + l * 64;
#pragma unroll 9
            for (int q = 0; q < W / 2; q++) o[(2 * q) * 64] = acc0 + acc1 * q;
        }
    }
}

int main() {
    const long nf = 262144, ntiles = nf / 64;
    float *in, *out;
    hipMalloc(&in, (size_t)ntiles * E * 64 * 4);
    hipMalloc(&out, (size_t)ntiles * W * 64 * 4);
    hipMemset(in, 0, (size_t)ntiles * E * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[4] = {"lane/filter [e][64]      ", "pair 2f+l on [e][64]     ", "pair 2f+l on [e/2][32][2]", "half 32l+f on [e][64]    "};
    for (int m = 0; m < 4; m++)
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; i++) {
                const long waves = m == 0 ? ntiles : 2 * ntiles;
                const dim3 g((unsigned)((waves + 3) / 4)), b(256);
                if (m == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, in, out, nf);
                if (m == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, in, out, nf);
                if (m == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, in, out, nf);
                if (m == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, in, out, nf);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            printf("%s rep %d: %6.1f us per launch, %.2f TB/s (read 357 MB + write 94 MB)\n", names[m], rep, ms * 1e3,
                   (double)nf * (E + W) * 4 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
