// kb_vanilla_reg.h -- the register-resident Vanilla step kernel (vanilla.go:128-220) and its launch
// helper, shared by the translation units that instantiate it for different shapes
// (kb_vanilla.hip: the benchmark shapes in fp64 and fp32 incl. the time-fused variant;
//  kb_vanilla_shapes.hip: the shapes of the reference's own examples and tests, fp64).
#pragma once
#include <type_traits>

#include "kb_internal.h"

namespace kb {

// waves (= tiles) per workgroup of the register kernel: 2 measured 1-1.5 % faster than 4 or 1 and 10 % faster than 8
// at 1M filters (smaller groups retire and refill more evenly)
#ifndef KB_VANILLA_WPB
#define KB_VANILLA_WPB 2
#endif

// Addressing: every block pointer below is WAVE-UNIFORM (tile index through readfirstlane), run-time
// field offsets are folded into that uniform base and the lane enters as a 32-bit offset, so the
// per-element offsets become instruction immediates: ~25 address computations per step instead of ~100.
template <typename T>
struct TilePtr {
    T *base;        // uniform: start of this wave's tile (or of one field inside it)
    unsigned lane;  // 0..63
    __device__ __forceinline__ T ld(int e) const { return (base + e * KB_TILE)[lane]; }
    __device__ __forceinline__ T ldnt(int e) const { return __builtin_nontemporal_load(base + e * KB_TILE + lane); }
    __device__ __forceinline__ void st(int e, T v) const { (base + e * KB_TILE)[lane] = v; }
    __device__ __forceinline__ void stnt(int e, T v) const { __builtin_nontemporal_store(v, base + e * KB_TILE + lane); }
    __device__ __forceinline__ TilePtr field(int first_elem) const { return TilePtr{base + (int64_t)first_elem * KB_TILE, lane}; }
};
template <typename T>
__device__ __forceinline__ T ldt(const TilePtr<T> &p, int e) { return p.ld(e); }
template <typename T>
__device__ __forceinline__ T ldt(const TilePtr<const T> &p, int e) { return p.ld(e); }
template <typename T>
__device__ __forceinline__ void stt(const TilePtr<T> &p, int e, T v) { p.st(e, v); }
// Streaming (read-once) operands -- the per-filter model F/H/Q/R/G and the measurements -- are
// loaded non-temporally so that they do not displace the state block (x, P: re-read and
// re-written every step, 226 MB at 1M filters) from the 256 MiB Infinity Cache.  Measured on
// MI355X with the arithmetic removed: 0.208 ms -> 0.153 ms per 1M-filter step.
template <typename T>
__device__ __forceinline__ T ldnt(const TilePtr<const T> &p, int e) { return p.ldnt(e); }
// Model operand of a kernel instantiated for SHARED batches (ONE model for every filter: StepArgs::mo_ts == 0): every lane reads
// lane 0's copy in tile 0's block.  The address is wave-uniform and nothing is stored before the last of these loads, so the
// compiler issues them as SCALAR loads (s_load, 92 of them at 6/3) and the model lives in SGPRs -- scalar operands of the FMAs --
// instead of 168 VGPRs: 238 -> 190 VGPRs, no model traffic beyond the scalar cache.  1M filters x 6/3: 98-100 us per step; as
// plain vector loads of the lane's own copy (the block stays in the L2) 105-110 us; with the streaming hint of the per-filter
// kernels 137 us; per-filter models 166 us.  (A third wave per SIMD -- a 168-register cap -- spills 36 registers: 125-140 us.)
template <bool SHARED, typename T>
__device__ __forceinline__ T ldm(const TilePtr<const T> &p, int e) {
    if constexpr (SHARED) return (p.base + e * KB_TILE)[0];   // lane 0's copy for every lane: a wave-uniform address
    else return p.ldnt(e);
}
template <typename T>
__device__ __forceinline__ T ldnt_at(const T *p) { return __builtin_nontemporal_load(p); }
// write-once Estimate extras (P-, K, innovation, yhat) are stored non-temporally for the same reason
template <typename T>
__device__ __forceinline__ void stnt(const TilePtr<T> &p, int e, T v) { p.stnt(e, v); }

// ---------------------------------------------------------------------------------
// register-resident kernel
// ---------------------------------------------------------------------------------
// PAD: the batch's run-time dimensions (a.n, a.p, a.m) may be smaller than the compile-time ones; the arithmetic runs
// on the padded operands -- zeros, and an identity block in R so that the innovation covariance stays invertible --
// which leaves every real entry of the result unchanged (the extra terms are exact zeros); only loads, stores and
// the condition-number test see the real sizes.  This is how shapes without an exact instantiation still get a
// register-resident kernel (kb_vanilla_pad.hip).
// NOISE: the batch's Noise is AWGN or BatchNoise (noise.go:67-164; wave-uniform choice at run time).  The three draws of a step
// keep the reference's call order -- Process(k) into x- (vanilla.go:146), Measurement(k) into yhat (:157), Process(k) again
// into x+ (:195) -- and its index k = kf.step of THIS filter (calls minus failed calls, kb_internal.h).  They are made where
// the register file has room: right in front of the innovation, after the gain, when F, P and Q are dead (x- is not used
// before that point, so the sums are the reference's, in the reference's order).
// NV standard normals of the draw (filter, kf.step, which) (kb_device.h: Philox4x32-10 + Box-Muller, two per block)
template <typename T, int NV>
__device__ __forceinline__ void draw_normals(const StepArgs &a, uint64_t gfi, uint32_t stepno, uint32_t which, T (&z)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; k += 2) {
        uint32_t r[4];
        Philox::gen(a.seed, gfi, stepno, ((uint32_t)(a.epoch * 4 + which) << 8) | (uint32_t)(k >> 1), r);
        double z0, z1;
        box_muller(r, z0, z1);
        z[k] = (T)z0;
        if (k + 1 < NV) z[k + 1] = (T)z1;
        __builtin_amdgcn_sched_barrier(0);   // one Box-Muller at a time: interleaved, their temporaries (log, sincospi in fp64) add up
    }
}
// w = L z with L = chol_L packed in the model block (L[i][k] at symi(k, i)); rows >= rv are padding
template <typename T, int NV>
__device__ __forceinline__ void chol_times(const TilePtr<const T> &moL, int rv, const T (&z)[NV], T (&w)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
        T s = T(0);
#pragma unroll
        for (int k = 0; k <= i; k++) s += ((i < rv) ? ldnt(moL, symi(k, i)) : T(0)) * z[k];
        w[i] = s;
    }
}
// chol_L of a packed symmetric positive definite matrix held in registers (A[i][j] at symi(i, j)), Dpotrf's order of
// operations (kb_dense.h cholesky_lower_rt); L[i][k], k <= i, at symi(k, i); rows / columns >= rv are padding and stay zero.
// The noise kernels form chol(Q) / chol(R) from the Q / R they have loaded anyway instead of reading the factors the
// constructor stored: 8 tri(n) bytes per filter-step less (168 of 1276 at n = 6), for ~60 FMAs.  kb_set_noise_kind has
// checked positive definiteness (noise.go:148-156 panics otherwise).
template <typename T, int NV>
__device__ __forceinline__ void chol_packed(const T (&A)[tri(NV)], int rv, T (&L)[tri(NV)]) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
        T ajj = A[symi(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) ajj -= L[symi(k, j)] * L[symi(k, j)];
        const bool real = j < rv;
        const T d = real ? sqrt(ajj) : T(0);
        const T inv = real ? T(1) / d : T(0);
        L[symi(j, j)] = d;
#pragma unroll
        for (int i = j + 1; i < NV; i++) {
            T sum = A[symi(j, i)];
#pragma unroll
            for (int k = 0; k < j; k++) sum -= L[symi(k, j)] * L[symi(k, i)];
            L[symi(j, i)] = (real && i < rv) ? sum * inv : T(0);
        }
    }
}
// w = L z, L packed as above
template <typename T, int NV>
__device__ __forceinline__ void tri_times(const T (&L)[tri(NV)], const T (&z)[NV], T (&w)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
        T s = T(0);
#pragma unroll
        for (int k = 0; k <= i; k++) s += L[symi(k, i)] * z[k];
        w[i] = s;
    }
}

template <typename T, int NS, int NM, int NC, bool FULL, bool PREDICT, bool FUSED, bool PAD = false, bool NOISE = false, bool SHARED = false>
__global__ void __launch_bounds__(KB_VANILLA_WPB * 64, (FUSED || (PAD && sizeof(T) * (NS * NS + NS * NM) > 8 * 56)) ? 1 : 2) vanilla_reg_kernel(const StepArgs a) {
    // FASTJ: the time-fused Noiseless kernel's own arithmetic (distributed Joseph form, Newton reciprocals).  The time-fused NOISE kernel
    // (round 5) keeps the per-step kernel's operations in its order instead: kb_update_steps_dev(T) with AWGN / BatchNoise is
    // bit-identical to T calls of kb_update_dev (tests/test_kinds_gpu.py), and the draws, not the Joseph form, are what it spends on.
    constexpr bool FASTJ = FUSED && !NOISE;
    constexpr int TR = tri(NS);
    constexpr int TM = tri(NM);
    const int rn = PAD ? a.n : NS, rp = PAD ? a.p : NM, rm = PAD ? a.m : NC;   // real sizes (compile-time constants unless PAD)
    const unsigned lane = threadIdx.x & 63u;
    const int64_t tile = (int64_t)blockIdx.x * KB_VANILLA_WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform
    if (tile >= a.ntiles) return;
    const bool active = tile * KB_TILE + lane < a.N;

    const TilePtr<T> st{(T *)a.state + tile * ((int64_t)KB_TILE * (rn + tri(rn))), lane};
    // (a batch with ONE model for all filters, mo_ts == 0: every lane reads lane 0's copy in tile 0 -- one 8-byte request per load
    // instead of a 512-byte row; the SHARED instantiations go further and make these scalar loads, see ldm)
    const TilePtr<const T> mo{(const T *)a.model + tile * a.mo_ts, a.mo_ts ? lane : 0u};
    const TilePtr<const T> moF = mo.field(a.L.mo_F), moH = mo.field(a.L.mo_H), moQ = mo.field(a.L.mo_Q), moR = mo.field(a.L.mo_R),
                           moG = mo.field(a.L.mo_G);
    const T *yp = (const T *)a.y + tile * a.y_ts + lane;
    const T *up = NC > 0 ? (const T *)a.u + tile * a.u_ts + lane : nullptr;

    // ---- state + transition model
    T x[NS], P[TR], F[NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) F[i * NS + j] = (i < rn && j < rn) ? ldm<SHARED>(moF, i * rn + j) : T(0);
    // Issue order is pinned with scheduling barriers, and it is "slowest first": the model and the measurement are HBM
    // streams (non-temporal), x and P are Infinity-Cache hits.  With F, Q, H, R, y requested before x and P every HBM request
    // of the wave is in flight as early as possible and the cache hits arrive right behind them: 163.5 us per 1M-filter step
    // against 171.9 us with x, P, F first (same box, three alternations; F alone moved forward: 171.6).
    __builtin_amdgcn_sched_barrier(0);
    [[maybe_unused]] T H[NM * NS], Q[TR], R[TM], G[NC > 0 ? NS * NC : 1], y0[NM];
#pragma unroll
    for (int j = 0; j < NS; j++)
#pragma unroll
        for (int i = 0; i <= j; i++) Q[symi(i, j)] = (j < rn) ? ldm<SHARED>(moQ, symi(i, j)) : T(0);
#pragma unroll
    for (int r = 0; r < NM; r++)
#pragma unroll
        for (int l = 0; l < NS; l++) H[r * NS + l] = (r < rp && l < rn) ? ldm<SHARED>(moH, r * rn + l) : T(0);
#pragma unroll
    for (int c = 0; c < NM; c++)
#pragma unroll
        for (int r = 0; r <= c; r++) R[symi(r, c)] = (c < rp) ? ldm<SHARED>(moR, symi(r, c)) : (r == c ? T(1) : T(0));
    if constexpr (NC > 0) {
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int c = 0; c < NC; c++) G[i * NC + c] = (i < rn && c < rm) ? ldm<SHARED>(moG, i * rm + c) : T(0);
    }
    if constexpr (!PREDICT) {
#pragma unroll
        for (int r = 0; r < NM; r++) y0[r] = (active && r < rp) ? ldnt_at(yp + (int64_t)r * a.y_es) : T(0);
    }
    // Cache policy of the state block (wave-uniform, chosen per batch at launch: StepArgs::stream_state).  A state block that fits the
    // 256 MiB Infinity Cache is read and written with the default policy and stays resident from step to step (with the model
    // streamed non-temporally around it); one that cannot fit anyway -- 4M filters: 906 MB -- is streamed non-temporally as well:
    // 740 us against 848 us per 4M-filter step, i.e. 6.26 TB/s = 0.995 of what a copy kernel achieves, where the resident policy
    // on a non-resident block thrashes (and the streaming policy at 1M filters costs 12 %: 189 against 168 us).
    auto load_state = [&](auto NT) {
#pragma unroll
        for (int i = 0; i < NS; i++) x[i] = (i < rn) ? (decltype(NT)::value ? st.ldnt(i) : st.ld(i)) : T(0);
#pragma unroll
        for (int j = 0; j < NS; j++)
#pragma unroll
            for (int i = 0; i <= j; i++)   // packed index does not depend on n
                P[symi(i, j)] = (j < rn) ? (decltype(NT)::value ? st.ldnt(rn + symi(i, j)) : st.ld(rn + symi(i, j))) : T(0);
    };
    KB_WITH_STATE_POLICY(a, load_state);
    __builtin_amdgcn_sched_barrier(0);

    unsigned err_acc = 0, nfail = 0;
    const int nsteps = FUSED ? a.nsteps : 1;
    [[maybe_unused]] const bool awgn = a.noise_kind == KB_NOISE_AWGN;
    for (int t = 0; t < nsteps; t++) {
        // ---- x- = F x [+ G u]
        T xm[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += F[i * NS + l] * x[l];
            xm[i] = s;
        }
        if constexpr (NC > 0) {
            T u[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) u[c] = (active && c < rm) ? ldnt_at(up + (int64_t)t * a.u_step + (int64_t)c * a.u_es) : T(0);
#pragma unroll
            for (int i = 0; i < NS; i++) {
                T s = T(0);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const T g = G[i * NC + c];
                    s += g * u[c];
                }
                xm[i] = xm[i] + s;
            }
        }
        // ---- P- = F P F^T + Q   (upper triangle; row i of F P, then dot with rows j >= i of F)
        T Pm[TR];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T fp[NS];
#pragma unroll
            for (int k = 0; k < NS; k++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += F[i * NS + l] * P[symi(l, k)];
                fp[k] = s;
            }
#pragma unroll
            for (int j = i; j < NS; j++) {
                T s = T(0);
#pragma unroll
                for (int k = 0; k < NS; k++) s += fp[k] * F[j * NS + k];
                Pm[symi(i, j)] = s + Q[symi(i, j)];
            }
        }
        // ---- yhat = H x_prev (previous posterior, vanilla.go:155-157)
        [[maybe_unused]] T yhat[NM];
        if constexpr (FULL) {
#pragma unroll
            for (int r = 0; r < NM; r++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += H[r * NS + l] * x[l];
                yhat[r] = s;
            }
        }
        [[maybe_unused]] T wpost[(NOISE && !PREDICT) ? NS : 1];
        if constexpr (NOISE) {
            // The draws sit between the time update and the gain: the point of the step with the fewest live values (P-, H, R, x-:
            // F, P and Q are dead, PH^T, S, K not yet formed).
            // pin (kb_device.h): the wave-uniform AWGN / BatchNoise branch below splits the kernel into basic blocks, and without it
            // LLVM sinks the whole time update BEHIND the branch, i.e. requests chol(Q) at the top of the kernel and carries it
            // through the register peak (measured: 272 B of scratch per lane)
#pragma unroll
            for (int e = 0; e < TR; e++) pin(Pm[e]);
#pragma unroll
            for (int i = 0; i < NS; i++) pin(xm[i]);
            if constexpr (FULL) {
#pragma unroll
                for (int r = 0; r < NM; r++) pin(yhat[r]);
            }
            __builtin_amdgcn_sched_barrier(0);   // the draws stay behind the gain (see the comment above draw_normals)
            // (global filter index and kf.step are formed HERE, not at the top: nothing of the noise path is alive in the load phase)
            const uint64_t gfi = (uint64_t)(a.first_filter + tile * KB_TILE) + lane;
            const uint32_t stepno = (uint32_t)a.step0 + (uint32_t)t - nfail - (active ? a.lag[tile * KB_TILE + lane] : 0u);   // kf.step of this filter
            if (awgn) {
                // all the normals first (only they are alive beside the filter's own values while the fp64 log / sincospi run),
                // then chol(Q) is formed ONCE (from Q, kept alive until here) and applied to both Process draws; w' waits in
                // 2 NS registers for x+
                T z0[NS];
                [[maybe_unused]] T z2[PREDICT ? 1 : NS];
                [[maybe_unused]] T z1[FULL ? NM : 1];
                draw_normals<T, NS>(a, gfi, stepno, 0u, z0);
                if constexpr (FULL) draw_normals<T, NM>(a, gfi, stepno, 1u, z1);
                if constexpr (!PREDICT) draw_normals<T, NS>(a, gfi, stepno, 2u, z2);
                T w[NS];
                {
                    T LQ[TR];
                    chol_packed<T, NS>(Q, rn, LQ);   // from the Q of the time update: no second stream (see chol_packed)
                    tri_times<T, NS>(LQ, z0, w);
                    if constexpr (!PREDICT) tri_times<T, NS>(LQ, z2, wpost);
                }
#pragma unroll
                for (int i = 0; i < NS; i++) xm[i] += w[i];                                        // Process(k), vanilla.go:146
                if constexpr (FULL) {
                    T v[NM], LR[TM];
                    chol_packed<T, NM>(R, rp, LR);
                    tri_times<T, NM>(LR, z1, v);
#pragma unroll
                    for (int r = 0; r < NM; r++) yhat[r] += v[r];                                  // Measurement(k), vanilla.go:157
                }
            } else {   // BatchNoise: the recorded vectors of step k (noise.go:72-86)
                const T *bp = (const T *)a.bn_proc + (int64_t)stepno * rn;
#pragma unroll
                for (int i = 0; i < NS; i++) {
                    const T w = (i < rn) ? bp[i] : T(0);
                    xm[i] += w;
                    if constexpr (!PREDICT) wpost[i] = w;
                }
                if constexpr (FULL) {
                    const T *bm = (const T *)a.bn_meas + (int64_t)stepno * rp;
#pragma unroll
                    for (int r = 0; r < NM; r++) yhat[r] += (r < rp) ? bm[r] : T(0);
                }
            }
            if constexpr (!PREDICT) {
#pragma unroll
                for (int i = 0; i < NS; i++) pin(wpost[i]);
            }
#pragma unroll
            for (int i = 0; i < NS; i++) pin(xm[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // FULL, one step per launch: yhat and (below) the innovation are complete long before they are stored, through the register
        // peak of the Joseph form.  They wait in LDS ([value][lane]: conflict-free, private to the lane) instead of in registers the
        // allocator would spill (36 B of scratch per lane otherwise).
        constexpr bool PARK = FULL && !FUSED && !PAD && sizeof(T) == 8;   // (fp32 and the padded one-wave-per-SIMD variants allocate better without)
        __shared__ T park_lds[PARK ? KB_VANILLA_WPB * 64 * 2 * NM : 1];
        [[maybe_unused]] volatile T *park = park_lds + (threadIdx.x >> 6) * (64 * 2 * NM) + lane;
        if constexpr (PARK) {
#pragma unroll
            for (int r = 0; r < NM; r++) park[r * 64] = yhat[r];
        }
        // ---- gain K = P- H^T (H P- H^T + R)^-1
        T PHt[NS * NM];
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int c = 0; c < NM; c++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += Pm[symi(i, l)] * H[c * NS + l];
                PHt[i * NM + c] = s;
            }
        T S[NM * NM], Si[NM * NM];
#pragma unroll
        for (int r = 0; r < NM; r++)
#pragma unroll
            for (int c = 0; c < NM; c++) {
                T s = T(0);
#pragma unroll
                for (int i = 0; i < NS; i++) s += H[r * NS + i] * PHt[i * NM + c];
                S[r * NM + c] = s + R[symi(r, c)];
            }
        unsigned err = inverse_lu<T, NM, FASTJ>(S, Si, rp) ? KB_ST_SINGULAR : 0u;
        T K[NS * NM];
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int c = 0; c < NM; c++) {
                T s = T(0);
#pragma unroll
                for (int k = 0; k < NM; k++) s += PHt[i * NM + k] * Si[k * NM + c];
                K[i * NM + c] = s;
            }

        T xn[NS], Pn[TR];
        [[maybe_unused]] T innov[NM];
        if constexpr (PREDICT) {
            // vanilla.go:170-179: estimate = {x-, yhat, 0, sym(P-), sym(P-), K}
#pragma unroll
            for (int i = 0; i < NS; i++) xn[i] = xm[i];
#pragma unroll
            for (int e = 0; e < TR; e++) Pn[e] = Pm[e];
#pragma unroll
            for (int r = 0; r < NM; r++) innov[r] = T(0);
        } else {
            // ---- innovation and state update
#pragma unroll
            for (int r = 0; r < NM; r++) {
                const T yv = (t == 0) ? y0[r] : ((active && r < rp) ? ldnt_at(yp + (int64_t)t * a.y_step + (int64_t)r * a.y_es) : T(0));
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += H[r * NS + l] * xm[l];
                innov[r] = yv - s;
            }
#pragma unroll
            for (int i = 0; i < NS; i++) {
                T s = T(0);
#pragma unroll
                for (int c = 0; c < NM; c++) s += K[i * NM + c] * innov[c];
                xn[i] = xm[i] + s;
            }
            if constexpr (NOISE) {
#pragma unroll
                for (int i = 0; i < NS; i++) xn[i] += wpost[i];   // vanilla.go:195: Process(k) a second time
            }
            if constexpr (PARK) {
#pragma unroll
                for (int r = 0; r < NM; r++) park[(NM + r) * 64] = innov[r];
            }
            if constexpr (FASTJ) {
                // ---- Joseph form with both multiplications by A = I - K H distributed (kb_vanilla_split.h has the argument):
                //   AP = P- - K (P- H^T)^T,   P+ = AP + (K R - AP H^T) K^T     [= A P- A^T + K R K^T, vanilla.go:197-205]
                // 333 FMAs where forming A, A P- and (A P-) A^T takes 567, and no 6 x 6 A in the register file.  The time-fused
                // kernel is bound by instruction issue (one wave per SIMD), so this is where it pays: 17.4 -> 19.5+ G filter-steps/s.
                // The per-step kernels are bound by HBM and keep the reference's order of operations: on a DEGENERATE problem (zero
                // noise matrices, the state pinned down exactly -- BatchNoise, noise.go:89-98) results are rounding noise amplified,
                // and only the same operations in the same order reproduce the reference's digits there.
                // AP H^T is formed from the COMPUTED AP: its rounding error (eps |P-|, as that of the reference's A P-) is
                // multiplied by A^T as in the reference's product.
#pragma unroll
                for (int i = 0; i < NS; i++) {   // row by row: only one row of AP is alive at a time
                    T ap[NS], v[NM];
#pragma unroll
                    for (int k = 0; k < NS; k++) {
                        T s = T(0);
#pragma unroll
                        for (int c = 0; c < NM; c++) s += K[i * NM + c] * PHt[k * NM + c];
                        ap[k] = Pm[symi(i, k)] - s;
                    }
#pragma unroll
                    for (int c = 0; c < NM; c++) {
                        T s = T(0);
#pragma unroll
                        for (int k = 0; k < NM; k++) s += K[i * NM + k] * R[symi(k, c)];
#pragma unroll
                        for (int k = 0; k < NS; k++) s -= ap[k] * H[c * NS + k];
                        v[c] = s;
                    }
#pragma unroll
                    for (int j = i; j < NS; j++) {
                        T s = ap[j];
#pragma unroll
                        for (int c = 0; c < NM; c++) s += v[c] * K[j * NM + c];
                        Pn[symi(i, j)] = s;
                    }
                }
            } else {
            // ---- Joseph form, upper triangle: P+ = K R K^T + A P- A^T,  A = I - K H
#pragma unroll
                for (int i = 0; i < NS; i++) {
                    T kr[NM];
#pragma unroll
                    for (int c = 0; c < NM; c++) {
                        T s = T(0);
#pragma unroll
                        for (int k = 0; k < NM; k++) s += K[i * NM + k] * R[symi(k, c)];
                        kr[c] = s;
                    }
#pragma unroll
                    for (int j = i; j < NS; j++) {
                        T s = T(0);
#pragma unroll
                        for (int c = 0; c < NM; c++) s += kr[c] * K[j * NM + c];
                        Pn[symi(i, j)] = s;
                    }
                }
                T A[NS * NS];
#pragma unroll
                for (int i = 0; i < NS; i++)
#pragma unroll
                    for (int j = 0; j < NS; j++) {
                        T s = T(0);
#pragma unroll
                        for (int c = 0; c < NM; c++) s += K[i * NM + c] * H[c * NS + j];
                        A[i * NS + j] = (i == j ? T(1) : T(0)) - s;
                    }
#pragma unroll
                for (int i = 0; i < NS; i++) {
                    T ap[NS];
#pragma unroll
                    for (int k = 0; k < NS; k++) {
                        T s = T(0);
#pragma unroll
                        for (int l = 0; l < NS; l++) s += A[i * NS + l] * Pm[symi(l, k)];
                        ap[k] = s;
                    }
#pragma unroll
                    for (int j = i; j < NS; j++) {
                        T s = T(0);
#pragma unroll
                        for (int k = 0; k < NS; k++) s += ap[k] * A[j * NS + k];
                        Pn[symi(i, j)] = s + Pn[symi(i, j)];
                    }
                }
            }
            }
        // ---- non-finite screen (stands in for AsSymDense's NaN-failing comparison)
        T chk = T(0);
#pragma unroll
        for (int i = 0; i < NS; i++) chk += xn[i] * T(0);
#pragma unroll
        for (int e = 0; e < TR; e++) chk += Pn[e] * T(0);
        if (chk != chk) err |= KB_ST_NONFINITE;
        // a failed step leaves (x, P) and kf.step as they were -- for THAT step only: the next one runs normally, as T calls of the
        // one-step kernel do (the fused loop used to freeze a filter at its first failure; the fused SquareRoot kernel did not)
        const bool ok = err == 0;
        err_acc |= err;
        nfail += ok ? 0u : 1u;   // vanilla.go:164-167, :207-215 return before kf.step++ (:218)

        if constexpr (FULL) {
            // Estimate extras of this step (only meaningful for the last fused step)
            if (active && ok) {
                const TilePtr<T> es0{(T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems), lane};
                const TilePtr<T> esP = es0.field(a.L.es_ppred), esK = es0.field(a.L.es_gain), esI = es0.field(a.L.es_innov), esY = es0.field(a.L.es_yhat);
#pragma unroll
                for (int j = 0; j < NS; j++)
#pragma unroll
                    for (int i = 0; i <= j; i++)
                        if (j < rn) stnt(esP, symi(i, j), Pm[symi(i, j)]);
#pragma unroll
                for (int i = 0; i < NS; i++)
#pragma unroll
                    for (int c = 0; c < NM; c++)
                        if (i < rn && c < rp) stnt(esK, i * a.pmax + c, K[i * NM + c]);
#pragma unroll
                for (int r = 0; r < NM; r++) {
                    if (r < rp) {
                        if constexpr (PARK) {
                            stnt(esI, r, (T)(PREDICT ? T(0) : park[(NM + r) * 64]));
                            stnt(esY, r, (T)park[r * 64]);
                        } else {
                            stnt(esI, r, innov[r]);
                            stnt(esY, r, yhat[r]);
                        }
                    }
                }
            }
        }
        if constexpr (FUSED) {
#pragma unroll
            for (int i = 0; i < NS; i++) x[i] = ok ? xn[i] : x[i];
#pragma unroll
            for (int e = 0; e < TR; e++) P[e] = ok ? Pn[e] : P[e];
        } else {
            if (active && ok) {
                auto store_state = [&](auto NT) {
#pragma unroll
                    for (int i = 0; i < NS; i++)
                        if (i < rn) { if constexpr (decltype(NT)::value) st.stnt(i, xn[i]); else st.st(i, xn[i]); }
#pragma unroll
                    for (int j = 0; j < NS; j++)
#pragma unroll
                        for (int i = 0; i <= j; i++)
                            if (j < rn) { if constexpr (decltype(NT)::value) st.stnt(rn + symi(i, j), Pn[symi(i, j)]); else st.st(rn + symi(i, j), Pn[symi(i, j)]); }
                };
                KB_WITH_STATE_POLICY(a, store_state);
            }
        }
    }
    if constexpr (FUSED) {
        if (active) {
#pragma unroll
            for (int i = 0; i < NS; i++)
                if (i < rn) stt(st, i, x[i]);
#pragma unroll
            for (int j = 0; j < NS; j++)
#pragma unroll
                for (int i = 0; i <= j; i++)
                    if (j < rn) stt(st, rn + symi(i, j), P[symi(i, j)]);
        }
    }
    if (active && err_acc) fail_step(a, tile * KB_TILE + lane, err_acc, nfail);
}

template <typename T, int NS, int NM, int NC, bool WITH_FUSED = true, bool NOISE = false, bool SHARED = false>
static inline bool try_reg(const Batch &b, const StepArgs &a, bool fused) {
    if (a.n != NS || a.p != NM || (a.need_ctrl ? a.m : 0) != NC) return false;
    if (SHARED && a.mo_ts != 0) return false;   // (the other instantiations are correct for shared batches too, only slower)
    if (fused && !WITH_FUSED) return false;
    if ((a.noise_kind != KB_NOISE_NOISELESS) != NOISE) return false;
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    const dim3 grid((unsigned)((a.ntiles + KB_VANILLA_WPB - 1) / KB_VANILLA_WPB)), block(KB_VANILLA_WPB * 64);
#define KB_GO(FULL_, PRED_, FUSED_) \
    KB_LAUNCH((vanilla_reg_kernel<T, NS, NM, NC, FULL_, PRED_, FUSED_, false, NOISE, SHARED>), grid, block, 0, b.stream, a)
    if constexpr (WITH_FUSED) {
        if (fused) {
            if (a.predict) { if (full) KB_GO(true, true, true); else KB_GO(false, true, true); }
            else           { if (full) KB_GO(true, false, true); else KB_GO(false, false, true); }
            return true;
        }
    }
    if (a.predict) { if (full) KB_GO(true, true, false); else KB_GO(false, true, false); }
    else           { if (full) KB_GO(true, false, false); else KB_GO(false, false, false); }
#undef KB_GO
    return true;
}

// Padded launch: any (n, p, m) with n <= NS, p <= NM, m <= NC (NC == 0 iff no control input), one step per launch.
template <typename T, int NS, int NM, int NC, bool NOISE = false, bool SHARED = false>
static inline bool try_pad(const Batch &b, const StepArgs &a) {
    const int m = a.need_ctrl ? a.m : 0;
    if (a.n > NS || a.p > NM || m > NC || (NC == 0) != (m == 0)) return false;
    if (SHARED && a.mo_ts != 0) return false;   // (the other instantiations are correct for shared batches too, only slower)
    if ((a.noise_kind != KB_NOISE_NOISELESS) != NOISE) return false;
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    const dim3 grid((unsigned)((a.ntiles + KB_VANILLA_WPB - 1) / KB_VANILLA_WPB)), block(KB_VANILLA_WPB * 64);
#define KB_GO(FULL_, PRED_) KB_LAUNCH((vanilla_reg_kernel<T, NS, NM, NC, FULL_, PRED_, false, true, NOISE, SHARED>), grid, block, 0, b.stream, a)
    if (a.predict) { if (full) KB_GO(true, true); else KB_GO(false, true); }
    else           { if (full) KB_GO(true, false); else KB_GO(false, false); }
#undef KB_GO
    return true;
}

// shapes instantiated in kb_vanilla_shapes.hip / kb_vanilla_pad*.hip
bool launch_vanilla_extra_shapes(const Batch &b, const StepArgs &a, bool fused);
bool launch_vanilla_padded(const Batch &b, const StepArgs &a);
bool launch_vanilla_padded8(const Batch &b, const StepArgs &a);
// AWGN / BatchNoise batches (kb_vanilla_noise.hip, kb_vanilla_noise_pad.hip): every shape up to n = 8, p = 4, m = 2, fp64
bool launch_vanilla_noise(const Batch &b, const StepArgs &a);
bool launch_vanilla_noise_fused(const Batch &b, const StepArgs &a);   // kb_vanilla_noise.hip: T steps in one launch, AWGN / BatchNoise, 6 / 3
bool vanilla_noise_fused_ok(const Batch &b, const StepArgs &a);      // kb_vanilla.hip
bool launch_vanilla_noise_padded(const Batch &b, const StepArgs &a);
// KB_FLAG_STRICT_SYMCHECK batches (kb_vanilla_strict.hip): both triangles, AsSymDense's test, the oracle's rounding; n <= 6, p <= 4, m <= 2
bool launch_vanilla_strict(const Batch &b, const StepArgs &a);
// batches whose filters all share ONE model (StepArgs::mo_ts == 0), fp64, one step per launch (kb_vanilla_shared.hip): the SHARED
// instantiations (model read with the default cache policy) of the exact and padded kernels, Noiseless and AWGN / BatchNoise
bool launch_vanilla_shared(const Batch &b, const StepArgs &a);
// one filter split over L lanes (kb_vanilla_split.h): n <= 12, p <= 8, m <= 2 (kb_vanilla_split12.hip), fp64, one step per launch
bool launch_vanilla_split12(const Batch &b, const StepArgs &a);
bool launch_vanilla_split16(const Batch &b, const StepArgs &a);   // 13..16 states: eight lanes per filter
bool launch_vanilla_split12_noise(const Batch &b, const StepArgs &a);   // kb_vanilla_split12n.hip: exact 12 / 6 / 0 with AWGN / BatchNoise
bool launch_vanilla_split12_plain(const Batch &b, const StepArgs &a);   // kb_vanilla_split12p.hip / 16p.hip: padded shapes, Noiseless, state only
bool launch_vanilla_split16_plain(const Batch &b, const StepArgs &a);

}  // namespace kb
