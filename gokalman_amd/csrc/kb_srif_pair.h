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
template <typename T, int NS, int NM, bool FULL, bool EXT, bool DENSE, bool PADM = false>
__device__ __forceinline__ void srif_pair_tile(const StepArgs &a, int64_t tile, int half, int lane, T *lds_lu) {
    static_assert(NS % 2 == 0 && NM % 2 == 0, "rows are split by parity");
    constexpr int COLS = NS + 1, HS = NS / 2, HM = NM / 2, SL = HS + HM, ROWE = NS * KB_TILE;
    typedef Panel<T, SL, HS> PN;
    typedef typename PN::V2 V2;
    // Addressing: every base below is wave-uniform (tile / half come from readfirstlane in the kernel), the per-lane part
    // is ONE 32-bit element offset (vf, or vrow for the own rows), so the loads and stores use the scalar-base +
    // 32-bit-vector-offset form: no 64-bit address pair per access (there are ~350 accesses per lane).
    const unsigned vf = (unsigned)lane & 31u;
    const bool is_hi = lane >= 32;
    const unsigned vrow = vf + (is_hi ? (unsigned)ROWE : 0u);   // own rows: element NS + 2 s NS + j from here is R[2 s + l][j]
    const unsigned vl = vf + (is_hi ? (unsigned)KB_TILE : 0u);  // own element of an (even, odd) pair of consecutive elements
    const int64_t first = tile * KB_TILE + half * 32;
    const int64_t fi = first + vf;
    const bool inb = fi < a.N;
    const unsigned vx = inb ? vf : 0u;   // caller's arrays end at N: lanes past it re-read the half-tile's first filter
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (NS + NS * NS)) + half * 32;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + half * 32;
    const T *ephi = EXT ? (const T *)a.ext_phi + first : nullptr;
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
    const unsigned vphi = EXT ? vx + (is_hi ? (unsigned)a.ext_ld : 0u) : vl;
    const unsigned bx = vx * (unsigned)sizeof(T), bphi = vphi * (unsigned)sizeof(T);

    unsigned err = 0;
    // own rows of the panel [[RBar bBar], [L Htilde, L y]] (the top part first holds the own rows of R), kept as COLUMN PAIRS
    // (2 c, 2 c + 1) plus the right-hand-side column: the Householder updates run on pairs (v_pk_fma_f32 in fp32: two
    // columns per instruction; in fp64 the same source compiles to two scalar FMAs)
    Panel<T, SL, HS> A;
    T xprev[NS];
    T pc[NS * HS];    // Phi, this half's columns: pc[r * HS + cs] = Phi[r][2 cs + l]; factorised in place
    auto load_phi = [&]() {
        UniformCursor<T> cur(EXT ? ephi : mo);   // walks Phi two elements at a time
#pragma unroll
        for (int r = 0; r < NS; r++)
#pragma unroll
            for (int cs = 0; cs < HS; cs++) {
                if constexpr (EXT) { pc[r * HS + cs] = cur.load_nt(bphi); cur.advance(2 * a.ext_ld); }
                else pc[r * HS + cs] = __builtin_nontemporal_load(mo + ((unsigned)((a.L.mo_F + r * NS + 2 * cs) * KB_TILE) + vphi));
            }
    };
    // ---- every operand is requested up front: ONE exposed memory latency per wave (the partner wave on the SIMD computes
    // meanwhile).  ~230 values in flight; the measurement operands (99) stay in registers until the factors of Phi have gone
    // to LDS, which is what the 256-register budget allows (whitening first would need 39 more accumulators on top).
    const int rp = PADM ? a.p : NM;
    T Hc[NM * HS], Lw[tri(NM)], yv[NM];   // Htilde, this half's columns: Hc[m * HS + cs] = Htilde[m][2 cs + l]
    [[maybe_unused]] T yreal[NM], yown[HM];
    auto load_meas = [&]() {
        {
            UniformCursor<T> cur(EXT ? eh : mo);   // walks Htilde two elements at a time
#pragma unroll
            for (int m = 0; m < NM; m++)
#pragma unroll
                for (int cs = 0; cs < HS; cs++) {
                    if (PADM && m >= rp) { Hc[m * HS + cs] = T(0); continue; }
                    if constexpr (EXT) { Hc[m * HS + cs] = cur.load_nt(bphi); cur.advance(2 * a.ext_ld); }
                    else Hc[m * HS + cs] = __builtin_nontemporal_load(mo + ((unsigned)((a.L.mo_H + m * NS + 2 * cs) * KB_TILE) + vphi));
                }
        }
#pragma unroll
        for (int c = 0; c < NM; c++)
#pragma unroll
            for (int m = 0; m <= c; m++)   // QUIRK srif.go:48: chol_L(R), not its inverse
                Lw[symi(m, c)] = (PADM && c >= rp) ? (m == c ? T(1) : T(0)) : ld_mo(a.L.mo_LR + symi(m, c));
        {
            UniformCursor<T> cr(yr), cc(yc);
#pragma unroll
            for (int r = 0; r < NM; r++) {
                if (PADM && r >= rp) { yv[r] = T(0); if constexpr (FULL) yreal[r] = T(0); continue; }
                const T re = cr.load_nt(bx), co = cc.load_nt(bx);
                cr.advance(a.y_es); cc.advance(a.y2_es);
                yv[r] = re - co;   // srif.go:143-144
                if constexpr (FULL) yreal[r] = re;
            }
        }
    };
    // fp64 (one wave per SIMD, nothing else hides its latency): the largest HBM stream, Phi, is requested first -- 3.5 % faster
    // than with it last; in fp32 the other order is 1 % ahead (b and R, the Infinity-Cache hits State(prev) starts from, arrive earlier)
    constexpr bool PHI_FIRST = sizeof(T) == 8 && !DENSE;
    // fp64 steady state: the measurement operands (63 values = 126 registers) are requested only when Phi's registers are free (its
    // factors are in LDS) and whitened after the RBar solves, whose ~25 us hide that second latency: 1143 -> 986 AGPR copies in the
    // code, 200 -> 191 us per 256k-filter step (profiles/NOTES.md).  In fp32 (two waves per SIMD, no AGPRs) the same order gains nothing.
    constexpr bool MEAS_LATE = sizeof(T) == 8 && !DENSE;
    if constexpr (PHI_FIRST) load_phi();
    if constexpr (!MEAS_LATE) load_meas();
    [[maybe_unused]] T bown[HS];   // b of the own rows
    if constexpr (DENSE) {
#pragma unroll
        for (int i = 0; i < NS; i++) xprev[i] = ld_st(i);   // b, in both halves
    } else {
#pragma unroll
        for (int s = 0; s < HS; s++) bown[s] = st[(unsigned)(2 * s * KB_TILE) + vl];
    }
    [[maybe_unused]] T Rw[DENSE ? NS * NS : 1];
    if constexpr (DENSE) {
#pragma unroll
        for (int e = 0; e < NS * NS; e++) Rw[e] = ld_st(NS + e);
    } else {
#pragma unroll
        for (int s = 0; s < HS; s++)
#pragma unroll
            for (int j = 0; j < NS; j++) A.set(s, j, j >= 2 * s ? ld_row(NS + 2 * s * NS + j) : T(0));   // (2 s + 1, 2 s) is a stored zero
        if constexpr (!PHI_FIRST) load_phi();
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- State(prev) = R^-1 b (srif.go:223-234) -------------------------------------------------------------------
    if constexpr (DENSE) {
        if (lu_solve_inplace<T, NS, 1>(Rw, xprev)) err |= KB_ST_SINGULAR;   // both halves, redundantly
        asm volatile("" ::: "memory");   // (cold path) the own rows are read after the 144 registers of Rw are free
#pragma unroll
        for (int s = 0; s < HS; s++)
#pragma unroll
            for (int j = 0; j < NS; j++) A.set(s, j, ld_row(NS + 2 * s * NS + j));
        load_phi();
    } else {
        // back substitution, column-oriented: as soon as x_i exists (in the half that owns row i) it is handed to the other
        // half and every remaining own row subtracts its R[r][i] x_i -- the critical path per component is one multiply, one
        // exchange and one FMA; the reciprocals of the diagonal are formed up front, off that path
        T rinv[HS], acc[HS];
#pragma unroll
        for (int s = 0; s < HS; s++) {
            const T d = is_hi ? A.get(s, 2 * s + 1) : A.get(s, 2 * s);   // R[2 s + l][2 s + l]
            if (d == T(0)) err |= KB_ST_SINGULAR;
            rinv[s] = recip(d);
            acc[s] = bown[s];
        }
#pragma unroll
        for (int i = NS - 1; i >= 0; i--) {
            const int si = i / 2, own = i % 2;
            xprev[i] = from_half(acc[si] * rinv[si], own);   // the other half's product (its row 2 si + 1 - own) is discarded
#pragma unroll
            for (int s = 0; s <= si; s++) acc[s] -= A.get(s, i) * xprev[i];   // rows 2 s + l < i; the structural zeros contribute 0
        }
    }
    KB_SRIF_STOP_AT(1, {
        for (int i = 0; i < NS; i++) sink__ += xprev[i];
        for (int e = 0; e < NS * HS; e++) sink__ += pc[e];
        if constexpr (!MEAS_LATE) { for (int e = 0; e < NM * HS; e++) sink__ += Hc[e]; for (int e = 0; e < tri(NM); e++) sink__ += Lw[e]; for (int e = 0; e < NM; e++) sink__ += yv[e]; }
    })
    // ---- xBar = Phi State(prev) (srif.go:118): each half sums over its columns ------------------------------------------
    T xbar[NS];
    {
        T xs[HS];
#pragma unroll
        for (int cs = 0; cs < HS; cs++) xs[cs] = is_hi ? xprev[2 * cs + 1] : xprev[2 * cs];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T part = T(0);
#pragma unroll
            for (int cs = 0; cs < HS; cs++) part += pc[i * HS + cs] * xs[cs];
            xbar[i] = allsum(part);
        }
    }
    // ---- P Phi = L U (srif.go:111-114's Inverse = Dgetrf + ...), the columns split over the halves: the half that owns
    // column j searches the pivot and forms the multipliers, the other half receives them (one exchange each) and both
    // update their own columns.  nibble k of perm = original index of the row now in position k.
    uint64_t perm = 0xFEDCBA9876543210ull;
#pragma unroll
    for (int j = 0; j < NS; j++) {
        const int cj = j / 2, oj = j % 2;
        const bool owner = is_hi == (oj == 1);
        bool need = false;   // some row below has a larger entry in column j: the bubble below would exchange at least once
#pragma unroll
        for (int r = j + 1; r < NS; r++) need = need || fabs(pc[r * HS + cj]) > fabs(pc[j * HS + cj]);
        if (__any(owner && need)) {   // wave-uniform and rare (one test per column): the exchange code only runs when some lane pivots
#pragma unroll
            for (int r = j + 1; r < NS; r++) {
                const bool sw = owner && fabs(pc[r * HS + cj]) > fabs(pc[j * HS + cj]);
                const bool s2 = from_half(sw ? 1u : 0u, oj) != 0u;
#pragma unroll
                for (int cs = 0; cs < HS; cs++) {   // whole rows: the L part moves with its row (LAPACK dlaswp)
                    const T t0 = pc[j * HS + cs], t1 = pc[r * HS + cs];
                    pc[j * HS + cs] = s2 ? t1 : t0;
                    pc[r * HS + cs] = s2 ? t0 : t1;
                }
                const uint64_t x = s2 ? (((perm >> (4 * j)) ^ (perm >> (4 * r))) & 15u) : 0u;
                perm ^= (x << (4 * j)) | (x << (4 * r));
            }
        }
        const T piv = pc[j * HS + cj];
        if (owner && piv == T(0)) err |= KB_ST_SINGULAR;
        const T rp = recip(piv);
        const T ujc = pc[j * HS + cj];          // upper half, j even: U[j][j + 1], still needed below
        if (owner) pc[j * HS + cj] = rp;        // the solves multiply by the reciprocal
#pragma unroll
        for (int r = j + 1; r < NS; r++) {
            const T lf = from_half(pc[r * HS + cj] * rp, oj);   // the multiplier, in both halves
            // slot cj: the owner keeps the multiplier (L); for an even j the upper half's column j + 1 is still active
            if (oj == 0) pc[r * HS + cj] = is_hi ? pc[r * HS + cj] - lf * ujc : lf;
            else pc[r * HS + cj] = is_hi ? lf : pc[r * HS + cj];
#pragma unroll
            for (int cs = cj + 1; cs < HS; cs++) pc[r * HS + cs] -= lf * pc[j * HS + cs];
        }
    }
    const bool anyswap = __any(perm != 0xFEDCBA9876543210ull);
    KB_SRIF_STOP_AT(2, {
        for (int i = 0; i < NS; i++) sink__ += xbar[i];
        for (int e = 0; e < NS * HS; e++) sink__ += pc[e];
        for (int s2 = 0; s2 < HS; s2++) for (int j = 0; j < NS; j++) sink__ += A.get(s2, j);
        if constexpr (!MEAS_LATE) { for (int e = 0; e < NM * HS; e++) sink__ += Hc[e]; for (int e = 0; e < tri(NM); e++) sink__ += Lw[e]; for (int e = 0; e < NM; e++) sink__ += yv[e]; }
    })
    // the factors go to LDS, [element][32 filters]: element (r, 2 cs + l) from this lane; the solves below read every
    // element from both halves (lanes f and 32 + f read the same word: a broadcast, no bank conflict)
#pragma unroll
    for (int r = 0; r < NS; r++)
#pragma unroll
        for (int cs = 0; cs < HS; cs++) lds_lu[(r * NS + 2 * cs) * 32 + lane] = pc[r * HS + cs];
    {   // a failure in either half fails the filter
        unsigned elo, ehi;
        halves(err, elo, ehi);
        err = elo | ehi;
    }
    const bool ok = inb && err == 0;   // failed: (b, R) stay as they are, srif.go:111-114 returns before any assignment
    if (err && inb && !is_hi) fail_step(a, fi, err);   // srif.go:112-114 returns before kf.step++
    if constexpr (DENSE) {
        // a filter that fails HERE keeps a dense R: its half-tile stays with this kernel (Batch::d_srif_dense; the steady-state kernel
        // never writes that word, so the two launches of one step cannot both take a half-tile) and the host keeps launching it
        // (Batch::srif_leftover) until a drained stream shows a launch in which nobody failed
        const bool left = __any(err != 0 && inb);
        if (lane == 0) {
            // every part bit of the word: fp32 at 14 / 16 states runs Predict() (and p = 7, 8) on kb_srif_split.h, which reads ONE BIT PER
            // PART of this word (a bare 1 left the part of slots 16..31 looking triangular to the next Predict())
            a.srif_dense[2 * tile + half] = left ? 0xFFFFFFFFu : 0u;
            if (left) __hip_atomic_store(a.srif_dense_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MEAS_LATE) { load_meas(); __builtin_amdgcn_sched_barrier(0); }
    // ---- whitened measurement rows (srif.go:146-148): [L Htilde | L y].  Each half forms ALL rows of L Htilde for ITS
    // columns (Htilde is held column-split like Phi: half the registers, half the loads), then one exchange per pair of
    // values turns columns-of-all-rows into all-columns-of-the-own-rows.
    auto whiten = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < HM; t++) {
#pragma unroll
        for (int cs = 0; cs < HS; cs++) {
            T w0 = T(0), w1 = T(0);   // rows 2 t and 2 t + 1, column 2 cs + l
#pragma unroll
            for (int m = 0; m <= 2 * t; m++) w0 += Lw[symi(m, 2 * t)] * Hc[m * HS + cs];
#pragma unroll
            for (int m = 0; m <= 2 * t + 1; m++) w1 += Lw[symi(m, 2 * t + 1)] * Hc[m * HS + cs];
            cross(w0, w1);   // lower half: row 2 t, columns 2 cs and 2 cs + 1; upper half: row 2 t + 1, the same columns
            A.set(HS + t, 2 * cs, w0);
            A.set(HS + t, 2 * cs + 1, w1);
        }
        T s0 = T(0), s1 = T(0);
#pragma unroll
        for (int m = 0; m <= 2 * t; m++) s0 += Lw[symi(m, 2 * t)] * yv[m];
#pragma unroll
        for (int m = 0; m <= 2 * t + 1; m++) s1 += Lw[symi(m, 2 * t + 1)] * yv[m];
        A.set(HS + t, NS, is_hi ? s1 : s0);
        if constexpr (FULL) yown[t] = is_hi ? yreal[2 * t + 1] : yreal[2 * t];
    }
    };
    if constexpr (!MEAS_LATE) whiten();

    if constexpr (!MEAS_LATE) {
        KB_SRIF_STOP_AT(3, {
            for (int i = 0; i < NS; i++) sink__ += xbar[i];
            for (int s2 = 0; s2 < SL; s2++) for (int j = 0; j <= NS; j++) sink__ += A.get(s2, j);
            for (int e = 0; e < NS * HS; e++) sink__ += pc[e];
        })
    }
    __builtin_amdgcn_sched_barrier(0);   // (!MEAS_LATE) the measurement operands are dead from here on
    // ---- RBar = R Phi^-1 (srif.go:115) for the own rows: z Phi = R[i,:], i.e. w U = r, v L = w, z[perm_k] = v_k --------
    // A[s][0..NS) is z for row 2 s + l; columns < 2 s are structural zeros (skipped) unless DENSE
    // two columns (2 c, 2 c + 1) at a time: the pair subtracts A[s][k2] (U[k2][2 c], U[k2][2 c + 1]) in one packed FMA
    auto lu2 = [&](int r, int c) { return V2{lds_lu[(r * NS + 2 * c) * 32 + vf], lds_lu[(r * NS + 2 * c + 1) * 32 + vf]}; };
    if constexpr (sizeof(T) == 8) {
    // (sfor: compile-time indices, no loop for the optimiser to unroll late -- the panel must be promoted to registers)
    sfor<0, HS>([&](auto C) __attribute__((always_inline)) {
        constexpr int c = C;
        sfor<0, 2 * c>([&](auto K2) __attribute__((always_inline)) {
            constexpr int k2 = K2;
            const V2 u = lu2(k2, c);
            sfor<0, HS>([&](auto S) __attribute__((always_inline)) {
                constexpr int s = S;
                if constexpr (DENSE || 2 * s <= k2) A.pair(s, c) = PN::fma2(-PN::splat(A.get(s, k2)), u, A.pair(s, c));
            });
        });
        const V2 d = lu2(2 * c, c);   // (1 / U[2c][2c], U[2c][2c + 1])
        const T r1 = lds_lu[((2 * c + 1) * NS + 2 * c + 1) * 32 + vf];
        sfor<0, HS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = S;
            if constexpr (DENSE || s <= c) {
                const T z0 = A.get(s, 2 * c) * d[0];
                A.set(s, 2 * c, z0);
                A.set(s, 2 * c + 1, (A.get(s, 2 * c + 1) - z0 * d[1]) * r1);
            }
        });
    });
    // v L = w from the last column back; within a pair the (2 c + 1) -> 2 c term comes last
    sfor<0, HS>([&](auto CR) __attribute__((always_inline)) {
        constexpr int c = HS - 1 - CR;
        sfor<2 * c + 2, NS>([&](auto K2) __attribute__((always_inline)) {
            constexpr int k2 = K2;
            const V2 lk = lu2(k2, c);
            sfor<0, HS>([&](auto S) __attribute__((always_inline)) {
                constexpr int s = S;
                A.pair(s, c) = PN::fma2(-PN::splat(A.get(s, k2)), lk, A.pair(s, c));
            });
        });
        const T l10 = lds_lu[((2 * c + 1) * NS + 2 * c) * 32 + vf];
        sfor<0, HS>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = S;
            A.set(s, 2 * c, A.get(s, 2 * c) - A.get(s, 2 * c + 1) * l10);
        });
    });
    } else {
        // fp32: column by column -- the pair form costs the allocator its slack at the 256-register cap (spills on the hot path)
#pragma unroll
        for (int j = 0; j < NS; j++) {
#pragma unroll
            for (int k2 = 0; k2 < j; k2++) {
                const T ukj = lds_lu[(k2 * NS + j) * 32 + vf];
#pragma unroll
                for (int s = 0; s < HS; s++)
                    if (DENSE || 2 * s <= k2) A.set(s, j, A.get(s, j) - A.get(s, k2) * ukj);
            }
            const T rjj = lds_lu[(j * NS + j) * 32 + vf];
#pragma unroll
            for (int s = 0; s < HS; s++)
                if (DENSE || 2 * s <= j) A.set(s, j, A.get(s, j) * rjj);
        }
#pragma unroll
        for (int j = NS - 2; j >= 0; j--) {
#pragma unroll
            for (int k2 = j + 1; k2 < NS; k2++) {
                const T lkj = lds_lu[(k2 * NS + j) * 32 + vf];
#pragma unroll
                for (int s = 0; s < HS; s++) A.set(s, j, A.get(s, j) - A.get(s, k2) * lkj);
            }
        }
    }
    // FULL: RBar leaves for the Estimate INSIDE each branch of the permutation test below.  Behind the join every entry of the panel is
    // a phi of the two branches, and a consumer of all 72 of them there costs ~50 registers at the kernel's peak (216 B of scratch in
    // fp32, 532 B in fp64, every y spilled as it arrives: 166 us against 88 us state-only) -- the Householder does not, it takes them
    // column by column.  Found by bisection (profiles/NOTES.md).
    auto store_rbar = [&]() __attribute__((always_inline)) {
        if constexpr (FULL) {
            if (ok && full_rt) {
#pragma unroll
                for (int s = 0; s < HS; s++)
#pragma unroll
                    for (int j = 0; j < NS; j++) __builtin_nontemporal_store(A.get(s, j), es + ((unsigned)((a.L.es_ppred + 2 * s * NS + j) * KB_TILE) + vrow));
            }
        }
    };
    // bBar = RBar xBar (srif.go:119), same products in pivoted order; then the row permutation is undone
    if (anyswap) {   // cold: some lane pivoted.  Register arrays cannot be indexed per lane: select chains
        T xp[NS];
#pragma unroll
        for (int r = 0; r < NS; r++) {
            const int pr = pnib(perm, r);
            T v = T(0);
#pragma unroll
            for (int c = 0; c < NS; c++) v = pr == c ? xbar[c] : v;
            xp[r] = v;
        }
#pragma unroll
        for (int s = 0; s < HS; s++) {
            T bb = T(0);
#pragma unroll
            for (int r = 0; r < NS; r++) bb += A.get(s, r) * xp[r];
            T row[NS];
#pragma unroll
            for (int c = 0; c < NS; c++) {
                T v = T(0);
#pragma unroll
                for (int r = 0; r < NS; r++) v = pnib(perm, r) == c ? A.get(s, r) : v;
                row[c] = v;
            }
#pragma unroll
            for (int c = 0; c < NS; c++) A.set(s, c, row[c]);
            A.set(s, NS, bb);
        }
        store_rbar();
    } else {
#pragma unroll
        for (int s = 0; s < HS; s++) {
            T bb = T(0);
#pragma unroll
            for (int r = 0; r < NS; r++) bb += A.get(s, r) * xbar[r];
            A.set(s, NS, bb);
        }
        store_rbar();
    }
    if constexpr (MEAS_LATE) { __builtin_amdgcn_sched_barrier(0); whiten(); }
    if constexpr (FULL) {
        if (ok && full_rt) {
#pragma unroll
            for (int t = 0; t < HM; t++) {
                if (PADM && 2 * t + (is_hi ? 1 : 0) >= rp) continue;   // (the padded row has no slot in the Estimate)
                __builtin_nontemporal_store(yown[t], es + ((unsigned)((a.L.es_yhat + 2 * t) * KB_TILE) + vl));
                __builtin_nontemporal_store(A.get(HS + t, NS), es + ((unsigned)((a.L.es_dobs + 2 * t) * KB_TILE) + vl));
            }
        }
    }

    KB_SRIF_STOP_AT(4, {
        for (int s2 = 0; s2 < SL; s2++) for (int j = 0; j <= NS; j++) sink__ += A.get(s2, j);
    })
    __builtin_amdgcn_sched_barrier(0);
    // ---- HouseholderTransf (helper.go:142-172) with the rows split over the halves ----------------------------------
    T chk = T(0);
#pragma unroll
    for (int k = 0; k < NS; k++) {
        const int sk = k / 2, lk = k % 2;
        // u_i = A[i][k] for the rows i >= k; in slot sk that is both halves when k is even, only the upper half when odd
        const int ck = k / 2;
        T colk[SL];   // column k of the own rows (register renaming)
#pragma unroll
        for (int s = sk; s < SL; s++) colk[s] = A.get(s, k);
        const T ask = colk[sk];
        T part = lk == 0 ? ask * ask : (is_hi ? ask * ask : T(0)), part2 = T(0);   // two chains: the sum is on the critical path of the step
#pragma unroll
        for (int s = sk + 1; s < SL; s++) {
            if ((s - sk) & 1) part2 += colk[s] * colk[s];
            else part += colk[s] * colk[s];
        }
        T sigma = allsum(part + part2);
        const T akk = from_half(ask, lk);
        const T sgn = (akk == T(0) || fabs(akk) <= T(1e-12)) ? T(1) : copysign(T(1), akk);   // helper.go:133-138 Sign
        sigma = root(sigma) * sgn;
        const T uk = akk + sigma;
        const T beta = recip(sigma * uk);
        const T usk = lk == 0 ? (is_hi ? ask : uk) : (is_hi ? uk : T(0));   // u of this lane's row in slot sk
        // The column pairs to the right of column k: both dot products in one accumulator pair, then ONE exchange for the
        // two columns: cross() leaves (first column's total | second column's total) in the (lower | upper) half after one
        // add, a second swap hands both totals to both halves -- 4 instructions for 2 columns; same sums as allsum()
#pragma unroll
        for (int c = 0; c < HS; c++) {
            if (c <= ck) continue;
            V2 pp = V2{T(0), T(0)};   // the rows below first: they do not wait for sigma
#pragma unroll
            for (int s = sk + 1; s < SL; s++) pp = PN::fma2(PN::splat(colk[s]), A.pair(s, c), pp);
            pp = PN::fma2(PN::splat(usk), A.pair(sk, c), pp);
            T p0 = pp[0], p1 = pp[1], g0, g1;
            cross(p0, p1);
            halves(p0 + p1, g0, g1);
            const V2 g = V2{g0 * beta, g1 * beta};
            A.pair(sk, c) = PN::fma2(-g, PN::splat(usk), A.pair(sk, c));
#pragma unroll
            for (int s = sk + 1; s < SL; s++) A.pair(s, c) = PN::fma2(-g, PN::splat(colk[s]), A.pair(s, c));
        }
        // the odd columns out: the right-hand side always, and column k + 1 (the other half of k's pair) when k is even
        if (lk == 0) {
            const int j = k + 1;
            T p0 = T(0), p1 = T(0);
#pragma unroll
            for (int s = sk + 1; s < SL; s++) {
                p0 += colk[s] * A.get(s, j);
                p1 += colk[s] * A.b[s];
            }
            p0 += usk * A.get(sk, j);
            p1 += usk * A.b[sk];
            T g0, g1;
            cross(p0, p1);
            halves(p0 + p1, g0, g1);
            g0 *= beta;
            g1 *= beta;
            A.set(sk, j, A.get(sk, j) - g0 * usk);
            A.b[sk] -= g1 * usk;
#pragma unroll
            for (int s = sk + 1; s < SL; s++) {
                A.set(s, j, A.get(s, j) - g0 * colk[s]);
                A.b[s] -= g1 * colk[s];
            }
        } else {
            T pj = T(0);
#pragma unroll
            for (int s = sk + 1; s < SL; s++) pj += colk[s] * A.b[s];
            pj += usk * A.b[sk];
            const T gamma = allsum(pj) * beta;
            A.b[sk] -= gamma * usk;
#pragma unroll
            for (int s = sk + 1; s < SL; s++) A.b[s] -= gamma * colk[s];
        }
        A.set(sk, k, lk == 0 ? (is_hi ? T(0) : -sigma) : (is_hi ? -sigma : ask));
        // row k is final: it leaves the register file from the half that owns it
#pragma unroll
        for (int j = k; j < COLS; j++) chk += A.get(sk, j) * T(0);
        if (ok && is_hi == (lk == 1)) {
            st[(unsigned)(k * KB_TILE) + vf] = A.b[sk];
#pragma unroll
            for (int j = k; j < NS; j++) st[(unsigned)((NS + 2 * sk * NS + j) * KB_TILE) + vrow] = A.get(sk, j);
        }
    }
    if constexpr (FULL) {
        if (ok && full_rt) {
#pragma unroll
            for (int t = 0; t < HM; t++) {
                if (PADM && 2 * t + (is_hi ? 1 : 0) >= rp) continue;
                __builtin_nontemporal_store(A.get(HS + t, NS), es + ((unsigned)((a.L.es_innov + 2 * t) * KB_TILE) + vl));
            }
        }
    }
    if constexpr (DENSE) {
        if (ok) {   // srif.go:334-337 zeroes the sub-columns; R was dense in memory
#pragma unroll
            for (int s = 0; s < HS; s++) {
#pragma unroll
                for (int j = 0; j < 2 * s; j++) st[(unsigned)((NS + 2 * s * NS + j) * KB_TILE) + vrow] = T(0);
                if (is_hi) st[(unsigned)((NS + 2 * s * NS + 2 * s) * KB_TILE) + vrow] = T(0);
            }
        }
    }
    // a non-finite result is stored as it is (helper.go:142-172 has no guard) and flagged
    if (ok && chk != chk) atomicOr(a.status + fi, (unsigned)KB_ST_NONFINITE);
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
