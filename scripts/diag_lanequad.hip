// Access-pattern model for a FOUR-lanes-per-filter SRIF kernel (16 filters per wave, lane = 16 l + f): 340 floats read and 90
// written per filter, 262144 filters.  The question: what do 64-byte segments cost?  Caller-planar arrays (Phi, Htilde, real,
// computed: 228 of the 340 words) are [element][N], so element e of a wave's 16 filters is 64 contiguous bytes and a wave-load
// touches four such segments of four different rows; the engine's own blocks (state, chol R: 112 words read, 90 written) can be
// laid out [e / 4][4][16] per quarter tile so that a wave-load is 256 contiguous bytes.
//   0  reference: lane = 32 l + f (two lanes per filter, today's kernel), [element][64] tiles, one wave per workgroup
//   1  quad, everything in 64-B segments on [element][64] tiles, workgroup = 4 waves = the 4 quarters of one tile
//   2  quad, everything in 64-B segments, one wave per workgroup, quarters of a tile in consecutive blocks (different XCDs)
//   3  quad, everything in 64-B segments, one wave per workgroup, quarters of a tile 8 blocks apart (same XCD)
//   4  quad, 228 words in 64-B segments + 112 words and all writes contiguous ([e/4][4][16]), quarters 8 blocks apart
//   5  quad, everything contiguous (upper bound)
// hipcc --offload-arch=gfx950 -O3 scripts/diag_lanequad.hip -o diag_lanequad
#include <hip/hip_runtime.h>
#include <stdio.h>

// -DPLAIN: the segment loads with the default cache policy instead of the streaming hint (round 5: with the hint a line whose halves /
// quarters are read by different workgroups leaves the L2 in between and comes from memory again, kb_srif_split.h)
#ifdef PLAIN
#define LDSEG(p) (*(p))
#else
#define LDSEG(p) __builtin_nontemporal_load(p)
#endif

constexpr int E = 340, W = 90, EP = 228;   // EP: words that arrive caller-planar

template <int MODE>
__global__ void __launch_bounds__(MODE == 1 ? 256 : 64) k(const float *__restrict__ in, float *__restrict__ out, long nfilters) {
    const int lane = threadIdx.x & 63;
    float acc0 = 0.f, acc1 = 0.f;
    if constexpr (MODE == 0) {
        const long wave = blockIdx.x;
        const long tile = wave >> 1;
        const int half = (int)(wave & 1);
        if (tile * 64 >= nfilters) return;
        const int f = lane & 31, l = lane >> 5;
        const float *p = in + tile * (long)(E * 64) + half * 32 + f + l * 64;
#pragma unroll 20
        for (int q = 0; q < E / 2; q += 2) { acc0 += __builtin_nontemporal_load(p + (2 * q) * 64); acc1 += __builtin_nontemporal_load(p + (2 * q + 2) * 64); }
        float *o = out + tile * (long)(W * 64) + half * 32 + f + l * 64;
#pragma unroll 9
        for (int q = 0; q < W / 2; q++) o[(2 * q) * 64] = acc0 + acc1 * q;
    } else {
        long tile; int quarter;
        if constexpr (MODE == 1) { tile = blockIdx.x; quarter = threadIdx.x >> 6; }
        else if constexpr (MODE == 2) { tile = blockIdx.x >> 2; quarter = blockIdx.x & 3; }
        else {   // blocks b, b + 8, b + 16, b + 24 of a group of 32 share a tile: same XCD under round-robin dispatch
            const long grp = blockIdx.x >> 5; const int r = blockIdx.x & 31;
            tile = grp * 8 + (r & 7); quarter = r >> 3;
        }
        if (tile * 64 >= nfilters) return;
        const int f = lane & 15, l = lane >> 4;
        const float *seg = in + tile * (long)(E * 64) + quarter * 16 + f + l * 64;          // element 4 g + l, 64-B segments
        const float *con = in + tile * (long)(E * 64) + quarter * (E * 16) + lane;          // [g][4][16] per quarter tile
        constexpr int NSEG = MODE == 5 ? 0 : (MODE == 4 ? EP / 4 : E / 4);
#pragma unroll 19
        for (int g = 0; g < NSEG; g++) acc0 += LDSEG(seg + (4 * g) * 64);
#pragma unroll 17
        for (int g = NSEG; g < E / 4; g++) acc1 += __builtin_nontemporal_load(con + g * 64);
        if constexpr (MODE >= 4) {
            float *o = out + tile * (long)(W * 64) + quarter * (W * 16) + lane;
#pragma unroll 11
            for (int g = 0; g < W / 4; g++) o[g * 64] = acc0 + acc1 * g;
            if (lane < 32) o[(W / 4) * 64] = acc0;
        } else {
            float *o = out + tile * (long)(W * 64) + quarter * 16 + f + l * 64;
#pragma unroll 11
            for (int g = 0; g < W / 4; g++) o[(4 * g) * 64] = acc0 + acc1 * g;
            if (l < 2) o[(4 * (W / 4)) * 64] = acc0;
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
    const char *names[6] = {"pair 32l+f [e][64], 1 wave/wg          ", "quad 64-B segments, wg = tile          ", "quad 64-B segments, quarters adjacent   ",
                            "quad 64-B segments, quarters same XCD   ", "quad 228 seg + 112 contiguous, same XCD ", "quad all contiguous                     "};
    for (int m = 0; m < 6; m++)
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; i++) {
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3((unsigned)(2 * ntiles)), dim3(64), 0, 0, in, out, nf);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3((unsigned)ntiles), dim3(256), 0, 0, in, out, nf);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3((unsigned)(4 * ntiles)), dim3(64), 0, 0, in, out, nf);
                if (m == 3) hipLaunchKernelGGL(k<3>, dim3((unsigned)(4 * ntiles)), dim3(64), 0, 0, in, out, nf);
                if (m == 4) hipLaunchKernelGGL(k<4>, dim3((unsigned)(4 * ntiles)), dim3(64), 0, 0, in, out, nf);
                if (m == 5) hipLaunchKernelGGL(k<5>, dim3((unsigned)(4 * ntiles)), dim3(64), 0, 0, in, out, nf);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            printf("%s rep %d: %6.1f us per launch, %.2f TB/s (read 357 MB + write 94 MB)\n", names[m], rep, ms * 1e3,
                   (double)nf * (E + W) * 4 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
