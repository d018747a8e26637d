// kb_srif_pair.h -- SRIF Update (srif.go:101-160, :298-340, helper.go:142-172) with TWO LANES PER FILTER.
//
// Why: with one filter per lane the 18 x 13 Householder panel alone is 234 values per lane, so the one-filter-per-lane
// kernels of round 1 ran at one wave per SIMD out of the 512-register budget: a lone wave issues one vector
// instruction every 4 cycles instead of 2, cannot overlap its own loads with its own arithmetic, and values beyond the
// 256 architectural VGPRs cost a copy per use.  Here a wave owns 32 filters: lane f (0..31) and lane 32 + f share filter
// f, the lower half of the wave holding the EVEN rows of every row-distributed matrix and the upper half the ODD rows.
// Per lane that is 9 x 13 = 117 panel values: the kernel fits 256 registers, two waves share a SIMD (one loads while the
// other computes), and nothing spills.
//
//   lane mapping   lane = 32 * l + f.  With the AoSoA-64 layout element e of 32 consecutive filters is one contiguous
//                  128-byte segment, so every wave-level load is two fully used segments (one per half) -- measured at the
//                  same 6.6 TB/s as the one-filter-per-lane stream, where the interleaved mapping lane = 2 f + l reads at
//                  5.2 TB/s on the same layout (scripts/diag_lanepair.hip).  The halves exchange values with
//                  v_permlane32_swap_b32 (gfx950): swap(x, x) leaves [x.lo | x.lo] and [x.hi | x.hi].
//   time update    x = R^-1 b by back substitution, one exchange per component (row i lives in half i % 2);
//                  Phi's COLUMNS are split the same way (72 registers per lane): xBar = Phi x is a partial sum per half + one
//                  exchange per row; in P Phi = L U the half that owns column j finds the pivot and forms the multipliers,
//                  the other half receives them (one exchange each), both update their own columns.  The factors go to LDS
//                  once per filter ([element][32 filters], 18 KB per wave in fp32, conflict-free, broadcast to both
//                  halves), and each half solves z Phi = R[i,:] for ITS six rows i, all six at once (one LDS read per
//                  factor element).
//                  RBar = R Phi^-1 therefore appears directly in the lanes that own those rows of the panel.
//   measurement    rows of [L Htilde | L y] are formed where they live; Householder with rows distributed over the halves:
//                  per column one partial dot product per half + one exchange.  Row k is final after step k and is stored
//                  from the half that owns it.
// Arithmetic as in the Predict() kernel of kb_srif_reg.hip (LU solves instead of inverse-then-multiply; exact singularity / non-finite flags);
// dot products that span both halves are summed as (even rows) + (odd rows), a rounding-level reordering.
// Failure semantics as everywhere (kb_srif_reg.hip header): a singular Phi / R skips this step for that filter only.
#pragma once
#include <type_traits>

#include "kb_internal.h"
#include "kb_static.h"

namespace kb {

// Diagnostic builds only (scripts/srif_phases.sh, profiles/<tag>/srif_phases.md): -DKB_SRIF_STOP=k ends the Update after phase k --
// 1 operands loaded + State(prev), 2 xBar + LU of Phi, 3 whitened measurement rows, 4 RBar and bBar -- with everything computed so
// far kept alive by a store that never happens; the differences between the variants' times are the phases' costs under load.
#ifndef KB_SRIF_STOP
#define KB_SRIF_STOP 0
#endif
#define KB_SRIF_STOP_AT(k, sum_expr)                                        \
    if constexpr (KB_SRIF_STOP == (k)) {                                    \
        T sink__ = T(0);                                                    \
        sum_expr;                                                           \
        if (sink__ == T(-1.2345e30)) st[vf] = sink__;                       \
        return;                                                             \
    }

// ---- exchange between lane f and lane 32 + f -------------------------------------------------------
__device__ __forceinline__ void halves(unsigned x, unsigned &lo, unsigned &hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // [x.lo | x.lo], [x.hi | x.hi]
    lo = r[0];
    hi = r[1];
}
__device__ __forceinline__ void halves(float x, float &lo, float &hi) {
    unsigned a, b;
    halves(__float_as_uint(x), a, b);
    lo = __uint_as_float(a);
    hi = __uint_as_float(b);
}
__device__ __forceinline__ void halves(double x, double &lo, double &hi) {
    unsigned al, bl, ah, bh;
    halves((unsigned)__double2loint(x), al, bl);
    halves((unsigned)__double2hiint(x), ah, bh);
    lo = __hiloint2double((int)ah, (int)al);
    hi = __hiloint2double((int)bh, (int)bl);
}
// y <- [y.lo | x.lo], x <- [y.hi | x.hi]: each half hands one value to the other and keeps one, in ONE instruction
__device__ __forceinline__ void cross(float &y, float &x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(x), false, false);
    y = __uint_as_float(r[0]);
    x = __uint_as_float(r[1]);
}
__device__ __forceinline__ void cross(double &y, double &x) {
    const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(y), (unsigned)__double2loint(x), false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(y), (unsigned)__double2hiint(x), false, false);
    y = __hiloint2double((int)rh[0], (int)rl[0]);
    x = __hiloint2double((int)rh[1], (int)rl[1]);
}
// sum over the two halves, identical in both (even-row part + odd-row part)
template <typename T>
__device__ __forceinline__ T allsum(T x) {
    T lo, hi;
    halves(x, lo, hi);
    return lo + hi;
}
// the value held by half `h` (compile-time), in both halves
template <typename T>
__device__ __forceinline__ T from_half(T x, int h) {
    T lo, hi;
    halves(x, lo, hi);
    return h ? hi : lo;
}

// sqrt: fp32 takes the hardware instruction (1 ulp) in place of the correctly rounded sequence the compiler expands sqrtf()
// into (15 instructions, once per Householder column); fp64 keeps the library expansion
__device__ __forceinline__ float root(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double root(double x) { return sqrt(x); }

__device__ __forceinline__ int pnib(uint64_t perm, int r) { return (int)((perm >> (4 * r)) & 15u); }

// SL rows x (2 HS + 1) columns in registers: HS column pairs per row and the last column apart.  All indices are compile-time
// constants once the loops are unrolled.
template <typename T, int SL, int HS>
struct Panel {
    typedef T V2 __attribute__((ext_vector_type(2)));
    V2 p[SL * HS];
    T b[SL];
    __device__ __forceinline__ T get(int s, int j) const { return j == 2 * HS ? b[s] : p[s * HS + j / 2][j & 1]; }
    __device__ __forceinline__ void set(int s, int j, T v) {
        if (j == 2 * HS) b[s] = v;
        else p[s * HS + j / 2][j & 1] = v;
    }
    __device__ __forceinline__ V2 &pair(int s, int c) { return p[s * HS + c]; }
    static __device__ __forceinline__ V2 splat(T u) { return (V2){u, u}; }
    static __device__ __forceinline__ V2 fma2(V2 a, V2 b, V2 c) { return __builtin_elementwise_fma(a, b, c); }
};

// One half-tile (32 filters) of the SRIF Update.  DENSE: R may be a full matrix for some filter of this half-tile (the
// Update right after a Predict(), or a filter that skipped such an Update): all of R is read, State(prev) is a pivoted LU
// solve, and the finished factor's lower triangle is zeroed in memory.  DENSE = false is the steady state (R upper
// triangular, structural zeros skipped at compile time).  The two variants are separate KERNELS (srif_pair_kernel,
// srif_pair_dense_kernel): inside one kernel the dense variant's 144-register copy of R set the allocation of both (104 B of
// scratch per lane in the code object although the steady-state path never touched it).
// PADM: the batch has p = NM - 1 measurements (p = 1 on a two-row, 3 on a four-row, 5 on a six-row instantiation): the missing row is
// a zero row of Htilde with a zero residual and a unit entry in chol(R) -- its whitened row is zero, the Householder steps add exact
// zeros for it, so every real entry of the result has the same bits as without the row; loads and Estimate stores skip it.
// FUSED (round 6, kb_update_nl_steps_dev): the caller loop `for k { kf.Prepare(Phi_k, Htilde_k); kf.Update(real_k, computed_k) }` inside one
// launch, steady state only (!DENSE, !FULL, EXT): step t reads the caller's arrays at t x their step strides, the own rows of (b, R) stay
// in the panel's registers from one step to the next (they are exactly where the next step would load them) and are still stored every
// step -- memory stays current, so a step that fails for some filter (its stores are predicated, as ever) makes the whole wave reload
// at the top of the next one; the same operations in the same order as T single launches: the same bits.
template <typename T, int NS, int NM, bool FULL, bool EXT, bool DENSE, bool PADM = false, bool FUSED = false>
__device__ __forceinline__ void srif_pair_tile(const StepArgs &a, int64_t tile, int half, int lane, T *lds_lu) {
    static_assert(NS % 2 == 0 && NM % 2 == 0, "rows are split by parity");
    static_assert(!FUSED || (EXT && !DENSE && !FULL && !PADM), "the time-fused variant: zero-copy operands, steady state, state only");
    constexpr int COLS = NS + 1, HS = NS / 2, HM = NM / 2, SL = HS + HM, ROWE = NS * KB_TILE;
    typedef Panel<T, SL, HS> PN;
    typedef typename PN::V2 V2;
    // Addressing: every base below is wave-uniform (tile / half come from readfirstlane in the kernel), the per-lane part
    // is ONE 32-bit element offset (vf, or vrow for the own rows), so the loads and stores use the scalar-base +
    // 32-bit-vector-offset form: no 64-bit address pair per access (there are ~350 accesses per lane).
    typedef std::conditional_t<FUSED, unsigned, const unsigned> lane_off;   // (laundered at the top of every FUSED step, below; otherwise constants)
    lane_off vf = (unsigned)lane & 31u;
    const bool is_hi = lane >= 32;
    lane_off vrow = vf + (is_hi ? (unsigned)ROWE : 0u);   // own rows: element NS + 2 s NS + j from here is R[2 s + l][j]
    lane_off vl = vf + (is_hi ? (unsigned)KB_TILE : 0u);  // own element of an (even, odd) pair of consecutive elements
    const int64_t first = tile * KB_TILE + half * 32;
    const int64_t fi = first + vf;
    const bool inb = fi < a.N;
    lane_off vx = inb ? vf : 0u;   // caller's arrays end at N: lanes past it re-read the half-tile's first filter
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (NS + NS * NS)) + half * 32;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + half * 32;
    const T *ephi = EXT ? (const T *)a.ext_phi + first : nullptr;   // (FUSED: advanced by the step strides at the end of every step)
    const T *eh = EXT ? (const T *)a.ext_h + first : nullptr;
    const T *yr = (const T *)a.y + tile * a.y_ts + half * 32;
    const T *yc = (const T *)a.y2 + tile * a.y2_ts + half * 32;
    // (beyond 12 states ONE instantiation serves batches with and without KB_FLAG_FULL_ESTIMATE -- srif_pair_launch: half the kernels of
    // the largest translation units for shapes no benchmark times -- so the stores are gated at run time as well)
    const bool full_rt = FULL && (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    T *es = full_rt ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + half * 32 : nullptr;
    auto ld_st = [&](int e) { return st[(unsigned)(e * KB_TILE) + vf]; };            // state element e of this lane's filter
    auto ld_row = [&](int e) { return st[(unsigned)(e * KB_TILE) + vrow]; };         // ... of the own row
    auto ld_mo = [&](int e) { return __builtin_nontemporal_load(mo + ((unsigned)(e * KB_TILE) + vf)); };
    // element e of Phi for the lower half, element e + 1 for the upper half (its column is the next one); the caller's planar
    // arrays have a run-time element stride, so their addresses are formed as (scalar pointer) + (32-bit lane byte offset)
    lane_off vphi = EXT ? vx + (is_hi ? (unsigned)a.ext_ld : 0u) : vl;
    lane_off bx = vx * (unsigned)sizeof(T), bphi = vphi * (unsigned)sizeof(T);

    // own rows of the panel [[RBar bBar], [L Htilde, L y]] (the top part first holds the own rows of R), kept as COLUMN PAIRS
    // (2 c, 2 c + 1) plus the right-hand-side column: the Householder updates run on pairs (v_pk_fma_f32 in fp32: two
    // columns per instruction; in fp64 the same source compiles to two scalar FMAs)
    Panel<T, SL, HS> A;
#ifndef KB_PAIR_FUSED_KEEP
#define KB_PAIR_FUSED_KEEP 1   // (0: diagnostic -- every fused step reloads its rows: they were written a step ago and sit in the L2)
#endif
    [[maybe_unused]] bool reload = true;   // FUSED: the own rows of (b, R) come from memory (first step, or some filter of the wave failed the last one)
    // The step itself is kb_srif_pair_step.inc, included TWICE: as it stands for the one-step kernels, and inside the loop of the FUSED one.
    // (A `for` around the one text changed the register allocation of the one-step instantiations although its trip count is a
    // compile-time 1 there -- 24 B of scratch in the fp64 12/6 kernel; a backward goto left the FUSED kernel with 316 B where the loop has 40.)
    if constexpr (FUSED) {
        for (int t = 0; t < a.nsteps; t++) {
            // Everything an address is formed from is made opaque per step: left alone, loop-invariant code motion hoists the ~350 element
            // offsets and the loads of chol(R) out of the loop and keeps them alive across it (530 B of scratch per lane)
            asm volatile("" : "+v"(vf), "+v"(vrow), "+v"(vl), "+v"(vx), "+v"(vphi), "+v"(bx), "+v"(bphi));
            unsigned long long ps = (unsigned long long)st, pm = (unsigned long long)mo;
            asm volatile("" : "+s"(ps), "+s"(pm));
            st = (T *)ps; mo = (const T *)pm;
#include "kb_srif_pair_step.inc"
            reload = !KB_PAIR_FUSED_KEEP || __any(err != 0);   // (a failed filter kept its old rows in memory only: the panel holds what the step made of them)
            ephi += a.ext_phi_step; eh += a.ext_h_step; yr += a.y_step; yc += a.y2_step;
        }
    } else {
#include "kb_srif_pair_step.inc"
    }
}

template <typename T, int NS>
constexpr int srif_pair_waves_per_simd() { return sizeof(T) * NS * NS * 32 * 4 * 2 <= 160 * 1024 ? 2 : 1; }

#ifndef KB_SRIF_FULL_ONE_WAVE
#define KB_SRIF_FULL_ONE_WAVE 0   // (1: the KB_FLAG_FULL_ESTIMATE variants at one wave per SIMD -- what helped while RBar was stored behind
                                  // the join of the permutation test, see store_rbar)
#endif
#ifndef KB_PAIR_WPB
#define KB_PAIR_WPB 1   // waves per workgroup.  They share nothing; with 4 per workgroup a finished wave's slot and LDS stay
                        // reserved until its three companions are done: 89.8 us against 86.7 us (fp32, 256k filters)
#endif
// Steady state: every R of the batch is upper triangular (Batch::srif_tri).  With a.srif_leftover the batch may still hold filters
// with a dense R (they failed the Update that followed a Predict()); the dense kernel marked their half-tiles in a.srif_dense, and
// those are left to srif_pair_dense_kernel, which the host launches right behind this kernel for as long as that can be the case.
template <typename T, int NS, int NM, bool FULL, bool EXT, bool PADM = false>
__global__ void __launch_bounds__(64 * KB_PAIR_WPB, ((KB_SRIF_FULL_ONE_WAVE && FULL) ? 1 : srif_pair_waves_per_simd<T, NS>())) srif_pair_kernel(const StepArgs a) {
    __shared__ T lds[KB_PAIR_WPB * NS * NS * 32];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: all bases become scalar
    const int64_t gw = (int64_t)blockIdx.x * KB_PAIR_WPB + wv;
    const int64_t tile = gw >> 1;
    const int half = (int)(gw & 1);
    const int64_t first = tile * KB_TILE + half * 32;
    if (first >= a.N) return;
    if (a.srif_leftover && a.srif_dense[gw] != 0u) return;   // srif_pair_dense_kernel's (launched right behind this one)
    srif_pair_tile<T, NS, NM, FULL, EXT, false, PADM>(a, tile, half, lane, lds + wv * (NS * NS * 32));
}

// The Update right after a Predict() (a.srif_tri == 0: every R is the dense RBar the Predict() kernel stored), or the half-tiles
// the steady-state kernel skipped (a.srif_tri != 0: only half-tiles marked in a.srif_dense).  One wave per SIMD: the
// register-resident copy of R needs the 512-register budget.
template <typename T, int NS, int NM, bool FULL, bool EXT, bool PADM = false>
__global__ void __launch_bounds__(64, 1) srif_pair_dense_kernel(const StepArgs a) {
    __shared__ T lds[NS * NS * 32];
    const int lane = threadIdx.x & 63;
    const int64_t gw = blockIdx.x;
    const int64_t tile = gw >> 1;
    const int half = (int)(gw & 1);
    const int64_t first = tile * KB_TILE + half * 32;
    if (first >= a.N) return;
    if (a.srif_tri && a.srif_dense[gw] == 0u) return;
    srif_pair_tile<T, NS, NM, FULL, EXT, true, PADM>(a, tile, half, lane, lds);
}

// kb_update_nl_steps_dev: a.nsteps Updates of the steady state in one launch (srif_pair_tile FUSED)
template <typename T, int NS, int NM>
__global__ void __launch_bounds__(64, (srif_pair_waves_per_simd<T, NS>())) srif_pair_fused_kernel(const StepArgs a) {
    __shared__ T lds[NS * NS * 32];
    const int lane = threadIdx.x & 63;
    const int64_t gw = blockIdx.x;
    const int64_t tile = gw >> 1;
    const int half = (int)(gw & 1);
    if (tile * KB_TILE + half * 32 >= a.N) return;
    srif_pair_tile<T, NS, NM, false, true, false, false, true>(a, tile, half, lane, lds);
}

template <typename T, int NS, int NM>
static bool srif_pair_launch_fused(const Batch &b, const StepArgs &a) {
    if (a.n != NS || a.p != NM || a.predict || !a.ext_phi || !a.srif_tri || a.srif_leftover || (a.flags & KB_FLAG_FULL_ESTIMATE)) return false;
    if (a.ext_ld >= (int64_t(1) << 28)) return false;
    KB_LAUNCH((srif_pair_fused_kernel<T, NS, NM>), dim3((unsigned)(2 * a.ntiles)), dim3(64), 0, b.stream, a);
    return true;
}

template <typename T, int NS, int NM, bool PADM = false>
static bool srif_pair_launch(const Batch &b, const StepArgs &a) {
    if (a.n != NS || (PADM ? (a.p != NM && a.p != NM - 1) : a.p != NM) || a.predict) return false;
    if (a.ext_phi && a.ext_ld >= (int64_t(1) << 28)) return false;   // the upper half's Phi offset (+ ld elements) is a 32-bit byte offset
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0, ext = a.ext_phi != nullptr;   // (NS > 12: always the FULL instantiation, its stores gated at run time)
    const dim3 grid((unsigned)((2 * a.ntiles + KB_PAIR_WPB - 1) / KB_PAIR_WPB)), block(64 * KB_PAIR_WPB);
    const dim3 dgrid((unsigned)(2 * a.ntiles)), dblock(64);
#define KB_P(F_, E_)                                                                                                                         \
    do {                                                                                                                                     \
        if (a.srif_tri) KB_LAUNCH((srif_pair_kernel<T, NS, NM, F_, E_, PADM>), grid, block, 0, b.stream, a);                             \
        if (!a.srif_tri || a.srif_leftover) KB_LAUNCH((srif_pair_dense_kernel<T, NS, NM, F_, E_, PADM>), dgrid, dblock, 0, b.stream, a); \
    } while (0)
    if constexpr (NS > 12) { if (ext) KB_P(true, true); else KB_P(true, false); }
    else {
        if (full) { if (ext) KB_P(true, true); else KB_P(true, false); }
        else      { if (ext) KB_P(false, true); else KB_P(false, false); }
    }
#undef KB_P
    return true;
}

}  // namespace kb
