// kb_srif_split.h -- SRIF Update / Predict (srif.go:101-160, :298-340, helper.go:142-172) in fp64 with ONE FILTER SPLIT OVER L LANES
// (L = 4 up to 12 states, L = 8 up to 16), the state dimension n at compile time (every n = 1 .. 16, odd n included: the padding to
// the next multiple of L is identity / zero in REGISTERS, only the real bytes move), p <= NM in {4, 6, 8} at run time.  Round 5:
// replaces, for fp64, the two-lane kernel everywhere but at 12/6 and 6/2 (beyond 12 states its panel sat partly in scratch at one wave
// per SIMD), the widened shadow copies of kb_srif_odd.hip for odd n, and the statement kernel for Predict() / p = 7, 8 at 13..16 states.
//
// Everything is distributed BY COLUMNS (lane = q (64 / L) + f as in kb_vanilla_split.h; lane q of a filter owns columns q, q + L, ...
// of R, Phi, Htilde and therefore of the Householder panel):
//
//   State(prev)   x = R^-1 b (srif.go:223-234).  Steady state (R upper triangular): column-oriented back substitution -- x_i is formed
//                 by the owner of column i from the lane sum of the partial products, then the owner adds R[:, i] x_i to its partials.
//                 After a Predict() R is the dense RBar: the part stages R in LDS and one lane per filter runs a pivoted LU solve there
//                 with rolled loops (cold: once per Predict).
//   xBar          Phi x: partial products over the own columns, then a reduce-scatter over the lanes: lane q ends with xBar of ITS columns
//                 (all that bBar = RBar xBar needs of it).
//   RBar          RBar = R Phi^-1 <=> Phi^T RBar^T = R^T: row j of [Phi^T | R^T] is (column j of Phi | column j of R) -- the lane's own
//                 data -- so a Gauss-Jordan elimination by ROWS with partial pivoting runs on it with the pivot row handed round
//                 through LDS (2 n - k values per step) and the elimination local.  Rows are never exchanged physically (the pivot
//                 row stays where it is, unnormalised, with its scale; LAPACK's row order is tracked as a position per row so that
//                 ties pick dgetf2's row), a filter that pivoted puts its rows back in order through LDS afterwards (cold).  The lane
//                 ends with ITS COLUMNS of RBar: exactly the panel's layout.  bBar = RBar xBar: partials + reduce-scatter, which leaves
//                 bBar distributed by rows -- where the right-hand side lives (below).
//   measurement   [L Htilde | L y] (srif.go:146-148; QUIRK :48: chol_L(R), not its inverse): own columns of Htilde, local.
//   Householder   (helper.go:142-172) by columns as in kb_squareroot_split.h: the ONE lane that owns column k forms sigma, u_k, beta --
//                 no sum over lanes; the squares in four partial sums -- and hands u and beta to the other lanes of the filter through
//                 LDS; every lane applies the reflection to its own columns right of k.  The right-hand side [bBar ; L y] is distributed
//                 by ROWS (lane q: rows q, q + L, ...): its dot product with u is one lane sum per step, its update 1 / L of the entries.
//
// Differences from the statement kernel (rounding level): solves instead of inverse-then-multiply, Newton reciprocals; only exact
// singularity is flagged (as every register SRIF path, kb_srif_reg.hip).  Failure semantics as everywhere: a singular Phi / R skips
// the step for that filter only.  A filter that fails the Update after a Predict() keeps a dense R: its part is marked (one bit per
// part in Batch::d_srif_dense's half-tile words) and takes the dense path until an Update of it succeeds (Batch::srif_leftover).
#pragma once
#include "kb_vanilla_split.h"

namespace kb {

#define KB_SB() __builtin_amdgcn_sched_barrier(0)
// The lane group holding the largest v wins, on a tie the smaller low byte of tag (LAPACK's first-largest in its row order): the
// same (v, tag) in all L lanes of a filter afterwards.
__device__ __forceinline__ void argmax_pick(double v0, unsigned t0, double v1, unsigned t1, double &v, unsigned &t) {
    const bool take1 = v1 > v0 || (v1 == v0 && (t1 & 0xffu) < (t0 & 0xffu));
    v = take1 ? v1 : v0;
    t = take1 ? t1 : t0;
}
__device__ __forceinline__ void argmax_pick(float v0, unsigned t0, float v1, unsigned t1, float &v, unsigned &t) {
    const bool take1 = v1 > v0 || (v1 == v0 && (t1 & 0xffu) < (t0 & 0xffu));
    v = take1 ? v1 : v0;
    t = take1 ? t1 : t0;
}
template <int L>
__device__ __forceinline__ void argmax_lanes(float &v, unsigned &tag) {
    {
        const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const auto t = __builtin_amdgcn_permlane32_swap(tag, tag, false, false);
        argmax_pick(__uint_as_float(x[0]), t[0], __uint_as_float(x[1]), t[1], v, tag);
    }
    if constexpr (L >= 4) {
        const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const auto t = __builtin_amdgcn_permlane16_swap(tag, tag, false, false);
        argmax_pick(__uint_as_float(x[0]), t[0], __uint_as_float(x[1]), t[1], v, tag);
    }
    if constexpr (L >= 8) {
        const float ov = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x128, 0xf, 0xf, false));
        const unsigned ot = (unsigned)__builtin_amdgcn_update_dpp(0, (int)tag, 0x128, 0xf, 0xf, false);
        const bool up = (threadIdx.x & 8u) != 0u;
        argmax_pick(up ? ov : v, up ? ot : tag, up ? v : ov, up ? tag : ot, v, tag);
    }
}
template <int L>
__device__ __forceinline__ void argmax_lanes(double &v, unsigned &tag) {
    {
        const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
        const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
        const auto t = __builtin_amdgcn_permlane32_swap(tag, tag, false, false);
        argmax_pick(__hiloint2double((int)h[0], (int)l[0]), t[0], __hiloint2double((int)h[1], (int)l[1]), t[1], v, tag);
    }
    if constexpr (L >= 4) {
        const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
        const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
        const auto t = __builtin_amdgcn_permlane16_swap(tag, tag, false, false);
        argmax_pick(__hiloint2double((int)h[0], (int)l[0]), t[0], __hiloint2double((int)h[1], (int)l[1]), t[1], v, tag);
    }
    if constexpr (L >= 8) {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x128, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x128, 0xf, 0xf, false);
        const unsigned ot = (unsigned)__builtin_amdgcn_update_dpp(0, (int)tag, 0x128, 0xf, 0xf, false);
        const double ov = __hiloint2double(hi, lo);
        const bool up = (threadIdx.x & 8u) != 0u;   // (lower lane's, upper lane's) in the same order on both sides
        argmax_pick(up ? ov : v, up ? ot : tag, up ? v : ov, up ? tag : ot, v, tag);
    }
}

// (rs32 / rs16 / rs8 and reduce_scatter: kb_vanilla_split.h)

// [R | b] staged for the dense State(prev) / the rows put back in order: NS^2 + NS; two pivot-row buffers: 4 NS; two reflector buffers
template <int NS, int NM, int L>
constexpr int srif_split_lds_elems() {
    int e = NS * NS + NS;
    if (4 * NS > e) e = 4 * NS;
    if (2 * (NS + NM + 1) > e) e = 2 * (NS + NM + 1);
    return e * (64 / L);
}

// N: the state dimension, exact (the panel is padded to NS = the next multiple of L in registers: identity / zero, never loaded or
// stored); NM >= p, the measurement dimension at run time.
template <typename T, int N, int NM, int L>
__device__ __forceinline__ void srif_split_part(const StepArgs &a, const int64_t gw, T *lds) {
    constexpr int NS = (N + L - 1) / L * L;
    static_assert(N >= 1 && NS <= 16 && NM <= 8, "columns are dealt out cyclically");
    constexpr int FPW = 64 / L, RP = NS / L, ROWS = NS + NM, rn = N;
    typedef __attribute__((address_space(1))) T *gptr;
    typedef const __attribute__((address_space(1))) T *cgptr;
    const int rp = a.p, rn_rt = a.n;   // (rn_rt == N)
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0, predict = a.predict != 0, ext = a.ext_phi != nullptr;
    const unsigned lane = threadIdx.x;
    const int q = (int)((lane / FPW) & (L - 1)), f = (int)(lane & (FPW - 1));
    const int64_t tile = gw / L;
    const int part = (int)(gw % L);
    if (tile * KB_TILE + part * FPW >= a.N) return;
    const int slot = part * FPW + f;
    const int64_t fi = tile * KB_TILE + slot;
    const bool active = fi < a.N;
    // which bit of the half-tile word (Batch::d_srif_dense) is this part's: 32 / FPW parts per half-tile
    const int half = (part * FPW) >> 5;
    const unsigned dbit = 1u << (((part * FPW) & 31) / FPW);
    uint32_t *const dword = a.srif_dense + (2 * tile + half);
    const bool dense = !a.srif_tri || (a.srif_leftover && (__builtin_amdgcn_readfirstlane((int)*dword) & (int)dbit) != 0);

    T *const st = (T *)a.state + tile * ((int64_t)KB_TILE * a.L.st_elems);
    const T *const mo = (const T *)a.model + tile * a.mo_ts;
    const unsigned us = (unsigned)slot, um = a.mo_ts ? (unsigned)slot : 0u;
    const unsigned uq = us + (unsigned)(q * KB_TILE), umq = um + (unsigned)(q * KB_TILE);
    // the caller's planar arrays (kb_prepare_dev) end at N: lanes past it re-read the part's first filter; 64-bit lane offsets
    const int64_t xoff = ext ? (int64_t)q * a.ext_ld + (tile * KB_TILE + (active ? slot : part * FPW)) : 0;
    // LDS layout: TWO consecutive elements per lane, element e at lp[PX(e)]: the pivot rows and the reflectors are read as contiguous runs --
    // ds_read_b128 (fp32: ds_read_b64) at twice the array rate of the pairs of narrow reads the compiler forms (kb_vanilla_split.h PAIRED; NOTES.md)
    // (A/B, NOTES.md: fp32 -4 ... -9 %, fp64 at eight lanes -1 ... -3 %; fp64 at four lanes nothing, and 8-16 B of scratch more at p = 7, 8: left as it was)
#ifdef KB_SRIF_SPLIT_UNPAIRED
    constexpr bool PAIRED = false;
#else
    constexpr bool PAIRED = sizeof(T) == 4 || L == 8;
#endif
    T *const lp = PAIRED ? lds + 2 * f : lds + f;
    auto PX = [](int e) constexpr -> int { return PAIRED ? (e >> 1) * (2 * FPW) + (e & 1) : e * FPW; };
    auto ep = [&](const T *ubase, int rt, int c) -> cgptr { return anchored(ubase, rt, c); };
    // Read-once streams (Phi, Htilde, chol R, observations): non-temporal at four lanes per filter, where a lane group reads a whole
    // 128-byte segment.  At eight lanes a group reads HALF a line and the part next door reads the other half a little later: with the
    // streaming hint the line is gone from the L2 by then and comes from memory twice (counters, 16/6: 5837 B read per filter-step against
    // 4296 B packed; two-wave workgroups over neighbouring parts did not help and cost 8 % -- NOTES.md), so there the default policy.
    // (fp32: four lanes per filter are 64-byte segments as well -- scripts/diag_lanequad.hip -DPLAIN: 6.2 TB/s with the default policy, 4.6 with the hint)
    constexpr bool PART_LINE = FPW * (int)sizeof(T) < 128;
    auto ldstream = [&](auto ptr) __attribute__((always_inline)) { if constexpr (PART_LINE) return *ptr; else return __builtin_nontemporal_load(ptr); };
    int jr[RP];
    bool colok[RP];
#pragma unroll
    for (int r = 0; r < RP; r++) { jr[r] = q + L * r; colok[r] = jr[r] < rn; }
    unsigned ulast[RP];   // the last REAL column of slot r, as a lane offset from column L r
#pragma unroll
    for (int r = 0; r < RP; r++) ulast[r] = us + (unsigned)(((L * r + L - 1 < N ? L - 1 : N - 1 - L * r) > 0 ? (L * r + L - 1 < N ? L - 1 : N - 1 - L * r) : 0) * KB_TILE);
    unsigned err = 0;

    // ---- phase 0: own columns of R (upper triangle only in the steady state) and of Phi, own b, own diagonal of R ----------------
    // (layout constants: st_vec = 0, st_mat = N, mo_F = 0 -- make_layout, kb_api.hip -- so every element index is an immediate)
    T Rc[RP][NS], Pc[RP][NS], bo[RP], dg[RP];
#pragma unroll
    for (int r = 0; r < RP; r++) {
#pragma unroll
        for (int i = 0; i < NS; i++) {
            if (i < N && L * r < N) {
                // steady state: rows below the slot's last column are never needed; rows inside it are needed by some lanes only
                const bool mine = colok[r] && (dense || i <= jr[r]);
                T v = T(0);
                // (a lane that does not need the entry re-reads one that another lane of the same instruction needs: no extra line)
                if (dense || i <= L * r + L - 1) v = *(ep(st, 0, N + i * N + L * r) + (mine ? uq : ulast[r]));
                Rc[r][i] = mine ? v : T(0);
            } else {
                Rc[r][i] = (!colok[r] && i == jr[r]) ? T(1) : T(0);
            }
        }
        if (L * r < N) {
            const T v = *(ep(st, 0, L * r) + (colok[r] ? uq : us));
            bo[r] = colok[r] ? v : T(0);
            const unsigned ud = us + (unsigned)(q * (N + 1) * KB_TILE);
            const T d = *(ep(st, 0, N + L * r * (N + 1)) + (colok[r] ? ud : us));
            dg[r] = colok[r] ? d : T(1);
        } else {
            bo[r] = T(0);
            dg[r] = T(1);
        }
    }
    auto load_phi = [&]() __attribute__((always_inline)) {
        // (ONE test of `ext` around the loops: inside them every load would carry its own branch and both address computations)
        if (ext) {
            const T *const xp = (const T *)a.ext_phi;
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int i = 0; i < NS; i++)
                    if (i < N && L * r < N) Pc[r][i] = ldstream(xp + ((int64_t)(i * N + L * r) * a.ext_ld + (colok[r] ? xoff : xoff - (int64_t)q * a.ext_ld)));
        } else {
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int i = 0; i < NS; i++)
                    if (i < N && L * r < N) Pc[r][i] = ldstream(ep(mo, 0, i * N + L * r) + (colok[r] ? umq : um));
        }
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int i = 0; i < NS; i++) {
                if (i < N && L * r < N) Pc[r][i] = colok[r] ? Pc[r][i] : T(0);
                else Pc[r][i] = (!colok[r] && i == jr[r]) ? T(1) : T(0);
            }
    };
    load_phi();
    KB_SB();

    // ---- phase 1: State(prev) = R^-1 b (srif.go:223-234), x_j in the owner of column j ------------------------------------------
    T xs[RP];
    if (dense) {
        // R is the dense RBar a Predict() stored (or this part holds a filter that failed the Update behind one): stage [R | b] in LDS,
        // one lane per filter solves with partial pivoting (dgetf2's choice), rolled loops over the real dimension
#pragma unroll
        for (int r = 0; r < RP; r++) {
#pragma unroll
            for (int i = 0; i < NS; i++) lp[PX(i * NS + jr[r])] = Rc[r][i];
            lp[PX(NS * NS + jr[r])] = bo[r];
        }
        wave_lds_fence();
        if (q == 0) {
            bool bad = false;
#pragma unroll 1
            for (int k = 0; k < rn; k++) {
                int piv = k;
                T best = fabs(lp[PX(k * NS + k)]);
#pragma unroll 1
                for (int i = k + 1; i < rn; i++) {
                    const T v = fabs(lp[PX(i * NS + k)]);
                    if (v > best) { best = v; piv = i; }
                }
                if (piv != k) {
#pragma unroll 1
                    for (int c = 0; c < rn; c++) {
                        const T t0 = lp[PX(k * NS + c)], t1 = lp[PX(piv * NS + c)];
                        lp[PX(k * NS + c)] = t1;
                        lp[PX(piv * NS + c)] = t0;
                    }
                    const T t0 = lp[PX(NS * NS + k)], t1 = lp[PX(NS * NS + piv)];
                    lp[PX(NS * NS + k)] = t1;
                    lp[PX(NS * NS + piv)] = t0;
                }
                const T pv = lp[PX(k * NS + k)];
                bad = bad || pv == T(0);
                const T ri = T(1) / pv;
#pragma unroll 1
                for (int i = k + 1; i < rn; i++) {
                    const T l = lp[PX(i * NS + k)] * ri;
#pragma unroll 1
                    for (int c = k + 1; c < rn; c++) lp[PX(i * NS + c)] -= l * lp[PX(k * NS + c)];
                    lp[PX(NS * NS + i)] -= l * lp[PX(NS * NS + k)];
                }
            }
#pragma unroll 1
            for (int i = rn - 1; i >= 0; i--) {
                T s = lp[PX(NS * NS + i)];
#pragma unroll 1
                for (int c = i + 1; c < rn; c++) s -= lp[PX(i * NS + c)] * lp[PX(NS * NS + c)];
                lp[PX(NS * NS + i)] = s / lp[PX(i * NS + i)];
            }
            if (bad) err |= KB_ST_SINGULAR;
        }
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < RP; r++) xs[r] = colok[r] ? lp[PX(NS * NS + jr[r])] : T(0);
        wave_lds_fence();
    } else {
        T rinv[RP], acc[NS];
#pragma unroll
        for (int r = 0; r < RP; r++) {
            if (dg[r] == T(0)) err |= KB_ST_SINGULAR;
            rinv[r] = recip(dg[r]);
            xs[r] = T(0);
        }
#pragma unroll
        for (int i = 0; i < NS; i++) acc[i] = T(0);
        sfor<0, NS>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = NS - 1 - I, r0 = i / L, q0 = i % L;
            if (i < rn && i < rn_rt) {   // (the padding contributes nothing; rn_rt: see the Householder loop)
                KB_SB();
                const T tot = sum_lanes<L>(acc[i]);   // sum_{j > i} R[i][j] x_j: the owners of the columns j > i have added their parts
                const bool own = q == q0;
                const T xi = (bo[r0] - tot) * rinv[r0];
                xs[r0] = own ? xi : xs[r0];
                const T xm = own ? xi : T(0);
#pragma unroll
                for (int i2 = 0; i2 < i; i2++) acc[i2] += Rc[r0][i2] * xm;
            }
        });
    }

    // ---- phase 2: xBar = Phi State(prev) (srif.go:118): partial sums over the own columns, lane sums two at a time ------------------
    T xbo[RP];   // xBar of the own columns (all bBar = RBar xBar needs of it)
    {
        T xb[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int r = 0; r < RP; r++) s += Pc[r][i] * xs[r];
            xb[i] = s;
        }
        reduce_scatter<L, NS, T>(xb, xbo);
#pragma unroll
        for (int r = 0; r < RP; r++) pin(xbo[r]);
    }
    KB_SB();

    // ---- phase 3: Gauss-Jordan on [Phi^T | R^T] by rows, partial pivoting (srif.go:111-115) ------------------------------------------
    // LDS: the normalised pivot row of the step, [0, NS) the Phi^T part (entries right of k), [NS, 2 NS) the R^T part
    bool used[RP];
    unsigned pos[RP], dest[RP];
    T scale[RP];
#pragma unroll
    for (int r = 0; r < RP; r++) { used[r] = false; pos[r] = (unsigned)jr[r]; dest[r] = (unsigned)jr[r]; scale[r] = T(1); }
    // Steady state, nobody has pivoted so far (wave-uniform `fast`): the owner of row k is known at compile time and hands its row
    // round SPECULATIVELY; every lane checks its own candidates against that pivot (a larger one anywhere: the step is redone the general
    // way below and the mode is left for good).  R^T is lower triangular and stays so while nobody pivots: the pivot row's right-hand
    // part ends at column k -- half the values, half the eliminations.
    bool fast = !dense;
    sfor<0, NS>([&](auto K) __attribute__((always_inline)) {
        constexpr int k = K, r0 = k / L, q0 = k % L;
        if (k < rn && k < rn_rt) {   // (the padding is an identity block, its steps change nothing; rn_rt: see the Householder loop)
            KB_SB();
            bool done = false;
            if (fast) {
                const bool own = q == q0;
                const T pv0 = Pc[r0][k];
                const T ri = recip(pv0);
                if (own) {
                    lp[PX(k)] = pv0;
#pragma unroll
                    for (int c = k + 1; c < NS; c++) lp[PX(c)] = Pc[r0][c] * ri;
#pragma unroll
                    for (int c = 0; c <= k; c++) lp[PX(NS + c)] = Rc[r0][c] * ri;
                }
                wave_lds_fence();
                const T pvb = fabs(lp[PX(k)]);
                bool need = false;   // a candidate row below with a larger entry (an equal one loses to row k: dgetf2 takes the first)
#pragma unroll
                for (int r = r0; r < RP; r++) need = need || ((r > r0 || q > q0) && colok[r] && fabs(Pc[r][k]) > pvb);
                if (!__any(need)) {
                    T pr[NS], prr[NS];
#pragma unroll
                    for (int c = k + 1; c < NS; c++) pr[c] = lp[PX(c)];
#pragma unroll
                    for (int c = 0; c <= k; c++) prr[c] = lp[PX(NS + c)];
                    if (own && pv0 == T(0)) err |= KB_ST_SINGULAR;
#pragma unroll
                    for (int r = 0; r < RP; r++) {
                        const T m = (r == r0 && own) ? T(0) : Pc[r][k];
#pragma unroll
                        for (int c = k + 1; c < NS; c++) Pc[r][c] -= m * pr[c];
#pragma unroll
                        for (int c = 0; c <= k; c++) Rc[r][c] -= m * prr[c];
                    }
                    scale[r0] = own ? ri : scale[r0];
                    used[r0] = used[r0] || own;
                    done = true;
                } else {
                    fast = false;
                }
                wave_lds_fence();
            }
            if (!done) {
                T bv = T(-1);
                unsigned btag = 0xffffffffu;
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    const T v = fabs(Pc[r][k]);
                    const bool take = !used[r] && colok[r] && (v > bv || (v == bv && pos[r] < (btag & 0xffu)));
                    bv = take ? v : bv;
                    btag = take ? (pos[r] | ((unsigned)jr[r] << 8)) : btag;
                }
                argmax_lanes<L>(bv, btag);
                const unsigned who = (btag >> 8) & 0xffu, wpos = btag & 0xffu;
                bool isp[RP];
#pragma unroll
                for (int r = 0; r < RP; r++) isp[r] = !used[r] && (unsigned)jr[r] == who;
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    if (isp[r]) {
                        const T pv = Pc[r][k];
                        if (pv == T(0)) err |= KB_ST_SINGULAR;
                        const T ri = recip(pv);
                        scale[r] = ri;
#pragma unroll
                        for (int c = k + 1; c < NS; c++) lp[PX(c)] = Pc[r][c] * ri;
#pragma unroll
                        for (int c = 0; c < NS; c++) lp[PX(NS + c)] = Rc[r][c] * ri;
                    }
                }
                wave_lds_fence();
                T pr[NS], prr[NS];
#pragma unroll
                for (int c = k + 1; c < NS; c++) pr[c] = lp[PX(c)];
#pragma unroll
                for (int c = 0; c < NS; c++) prr[c] = lp[PX(NS + c)];
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    const T m = isp[r] ? T(0) : Pc[r][k];
#pragma unroll
                    for (int c = k + 1; c < NS; c++) Pc[r][c] -= m * pr[c];
#pragma unroll
                    for (int c = 0; c < NS; c++) Rc[r][c] -= m * prr[c];
                }
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    // dgetf2 exchanges the rows at positions k and wpos: the row that sat at position k moves to the winner's position
                    pos[r] = (!used[r] && !isp[r] && pos[r] == (unsigned)k) ? wpos : pos[r];
                    dest[r] = isp[r] ? (unsigned)k : dest[r];
                    used[r] = used[r] || isp[r];
                }
                wave_lds_fence();
            }
        }
    });
    // the rows of Z = RBar^T: scale of the pivot row; a filter that pivoted holds them out of order
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int i = 0; i < NS; i++) Rc[r][i] *= scale[r];
    {
        bool moved = false;
#pragma unroll
        for (int r = 0; r < RP; r++) moved = moved || (colok[r] && dest[r] != (unsigned)jr[r]);
        if (__any(moved)) {   // cold
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int i = 0; i < NS; i++) lp[PX(dest[r] * NS + i)] = Rc[r][i];
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int i = 0; i < NS; i++) Rc[r][i] = lp[PX(jr[r] * NS + i)];
            wave_lds_fence();
        }
    }
    // Rc[r][i] = RBar[i][j_r] from here on.  bBar = RBar xBar (srif.go:119)
    // The right-hand side column [bBar ; L y] lives by ROWS: lane q carries rows q, q + L, ... (the state rows are exactly where the
    // reduce-scatter leaves bBar; 1 / L of the entries and of the arithmetic per lane, one lane sum per Householder step)
    constexpr int MP = (NM + L - 1) / L;
    T rv[RP + MP], bbo[RP];
    {
        T part[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int r = 0; r < RP; r++) s += Rc[r][i] * xbo[r];
            part[i] = s;
        }
        reduce_scatter<L, NS, T>(part, bbo);   // bBar[j_r] in the owner of column j_r
    }
    err = sum_lanes<L>(err);   // a failure anywhere fails the filter (flags: OR)
    const bool ok = active && err == 0;   // failed: (b, R) stay as they are, srif.go:111-114 returns before any assignment and before kf.step++
    if (err && active && q == 0) fail_step(a, fi, err);
    if (dense && !predict) {
        // this part stays with the dense path while one of its filters still holds a dense R; the host keeps the leftover mode on until
        // a drained stream shows a launch in which nobody failed (Batch::srif_leftover)
        const bool left = __any(err != 0 && active);
        if (lane == 0) {
            if (left) { atomicOr(dword, dbit); __hip_atomic_store(a.srif_dense_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            else atomicAnd(dword, ~dbit);
        }
    }
    T *const es = full ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) : nullptr;
    if (full && ok) {   // RBar is the Estimate's predicted matrix (srif.go:136, :153)
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int i = 0; i < NS; i++)
                if (colok[r] && i < rn) __builtin_nontemporal_store(Rc[r][i], (gptr)ep(es, 0, i * N + L * r) + uq);
    }
    if (predict) {   // srif.go:134-141: (bBar, RBar) become the estimate; zero observation vectors
        if (ok) {
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int i = 0; i < NS; i++)
                    if (colok[r] && i < rn) *((gptr)ep(st, 0, N + i * N + L * r) + uq) = Rc[r][i];
#pragma unroll
            for (int r = 0; r < RP; r++)
                if (colok[r]) *((gptr)ep(st, 0, L * r) + uq) = bbo[r];
            if (full) {
#pragma unroll
                for (int m = 0; m < 8; m++)   // (every row up to p: a Predict() runs on the p <= 4 instantiation whatever p is)
                    if (m < rp && q == m % L) {
                        __builtin_nontemporal_store(T(0), (gptr)ep(es, a.L.es_yhat, m) + us);
                        __builtin_nontemporal_store(T(0), (gptr)ep(es, a.L.es_dobs, m) + us);
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < RP; r++) rv[r] = bbo[r];
    KB_SB();

    // ---- phase 5: the whitened measurement rows (srif.go:143-148), own columns of Htilde ---------------------------------------------
    T W[RP][NM], real[NM], yw[NM];
    {
        T Hc[RP][NM], LR[tri(NM)], yv[NM];
        if (ext) {
            const T *const xh = (const T *)a.ext_h;
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int m = 0; m < NM; m++)
                    Hc[r][m] = (L * r < N && m < rp) ? ldstream(xh + ((int64_t)(m * N + L * r) * a.ext_ld + (colok[r] ? xoff : xoff - (int64_t)q * a.ext_ld))) : T(0);
        } else {
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int m = 0; m < NM; m++)
                    Hc[r][m] = (L * r < N && m < rp) ? ldstream(ep(mo, 0, N * N + m * N + L * r) + (colok[r] ? umq : um)) : T(0);
        }
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int m = 0; m < NM; m++) Hc[r][m] = colok[r] ? Hc[r][m] : T(0);
#pragma unroll
        for (int m = 0; m < NM; m++)
#pragma unroll
            for (int l = 0; l <= m; l++) LR[symi(l, m)] = m < rp ? ldstream(ep(mo, a.L.mo_LR, symi(l, m)) + um) : T(0);
        {
            const T *yr = (const T *)a.y + tile * a.y_ts, *yc = (const T *)a.y2 + tile * a.y2_ts;
#pragma unroll
            for (int m = 0; m < NM; m++) {
                const T re = m < rp ? ldstream(yr + ((int64_t)m * a.y_es + us)) : T(0);
                const T co = m < rp ? ldstream(yc + ((int64_t)m * a.y2_es + us)) : T(0);
                real[m] = re;
                yv[m] = re - co;   // srif.go:143-144
            }
        }
#pragma unroll
        for (int m = 0; m < NM; m++) {
#pragma unroll
            for (int r = 0; r < RP; r++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l <= m; l++) s += LR[symi(l, m)] * Hc[r][l];
                W[r][m] = s;
            }
            T s = T(0);
#pragma unroll
            for (int l = 0; l <= m; l++) s += LR[symi(l, m)] * yv[l];
            yw[m] = s;
        }
    }
#pragma unroll
    for (int t = 0; t < MP; t++) {   // the lane's own measurement rows q + L t
        T v = T(0);
#pragma unroll
        for (int qq = 0; qq < L; qq++)
            if (L * t + qq < NM) v = q == qq ? yw[L * t + qq] : v;
        rv[RP + t] = v;
    }
    if (full && ok) {
#pragma unroll
        for (int m = 0; m < NM; m++)
            if (m < rp && q == m % L) {
                __builtin_nontemporal_store(real[m], (gptr)ep(es, a.L.es_yhat, m) + us);
                __builtin_nontemporal_store(yw[m], (gptr)ep(es, a.L.es_dobs, m) + us);
            }
    }
    KB_SB();

    // ---- phase 6: HouseholderTransf (helper.go:142-172) by columns ----------------------------------------------------------------------
    // panel column j_r: rows [0, NS) = Rc[r][.], rows [NS, NS + NM) = W[r][.]; LDS: u_i at slot i, beta at slot ROWS
    auto A = [&](int r, int i) -> T & { return i < NS ? Rc[r][i] : W[r][i - NS]; };
    sfor<0, NS>([&](auto K) __attribute__((always_inline)) {
        constexpr int k = K, r0 = k / L, q0 = k % L;
        // (k < N always holds at run time: the opaque test ends the basic block.  As ONE block the sixteen steps are one selection DAG,
        // whose linearisation places pure arithmetic far from where it was written -- 2.3 KB of scratch per lane -- and sched_barrier only
        // binds the machine scheduler behind it)
        if (k < rn && k < rn_rt) {
            KB_SB();
            const bool own = q == q0;
            // (four partial sums: a chain of n + p - k dependent FMAs would be the longest thing in the step; the reference adds in row order)
            T s4[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int i = k; i < ROWS; i++) s4[(i - k) & 3] += A(r0, i) * A(r0, i);
            const T s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            const T akk = A(r0, k);
            const T sgn = (fabs(akk) <= T(1e-12)) ? T(1) : copysign(T(1), akk);   // helper.go:133-138 Sign
            const T sigma = sqrt(s) * sgn;   // (a reciprocal-square-root + Newton sequence, 11 instructions for ~25, changed nothing measurable: NOTES.md)
            const T uk = akk + sigma;
            const T beta = recip(sigma * uk);
            if (own) {
                lp[PX(k)] = uk;
                lp[PX(ROWS)] = beta;
#pragma unroll
                for (int i = k + 1; i < ROWS; i++) lp[PX(i)] = A(r0, i);
            }
            wave_lds_fence();
            T u[ROWS];
#pragma unroll
            for (int i = k; i < ROWS; i++) u[i] = lp[PX(i)];
            const T bt = lp[PX(ROWS)];
#pragma unroll
            for (int r = (q0 == L - 1 ? r0 + 1 : r0); r < RP; r++) {   // (q0 = L - 1: no column of slot r0 lies right of k)
                const bool right = r > r0 || q > q0;   // column j_r lies right of k
                T g2[2] = {T(0), T(0)};
#pragma unroll
                for (int i = k; i < ROWS; i++) g2[(i - k) & 1] += u[i] * A(r, i);
                T g = g2[0] + g2[1];
                g = right ? g * bt : T(0);
#pragma unroll
                for (int i = k; i < ROWS; i++) A(r, i) -= g * u[i];
            }
            {   // the right-hand side: own rows at or below k (u of the own rows straight from LDS: their index depends on the lane)
                T uo[RP + MP], gp = T(0);
#pragma unroll
                for (int t = 0; t < RP + MP; t++) {
                    constexpr int unused = 0; (void)unused;
                    const int base = t < RP ? L * t : NS + L * (t - RP);   // first row of slot t
                    if (base + L - 1 >= k && base < ROWS) {
                        const bool in = base + q >= k && base + q < ROWS;
                        const T v = lp[PX(in ? base + q : k)];
                        uo[t] = in ? v : T(0);
                        gp += uo[t] * rv[t];
                    } else {
                        uo[t] = T(0);
                    }
                }
                const T g = sum_lanes<L>(gp) * bt;
#pragma unroll
                for (int t = 0; t < RP + MP; t++) rv[t] -= g * uo[t];
            }
            A(r0, k) = own ? -sigma : A(r0, k);   // (helper.go:166-168 zeroes the sub-column: those registers are never read again)
            wave_lds_fence();
        }
    });

    // ---- results: b_k, the upper triangle of R_k (own columns); a dense R in memory gets its sub-columns zeroed (srif.go:334-337) ----
    T chk = T(0);
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int i = 0; i < NS; i++) chk += (i <= jr[r] ? Rc[r][i] : T(0)) * T(0);
#pragma unroll
    for (int t = 0; t < RP + MP; t++) chk += rv[t] * T(0);
    chk = sum_lanes<L>(chk);
    if (ok) {
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int i = 0; i < NS; i++)
                if (colok[r] && i < rn && (dense || i <= jr[r])) *((gptr)ep(st, 0, N + i * N + L * r) + uq) = i <= jr[r] ? Rc[r][i] : T(0);
#pragma unroll
        for (int r = 0; r < RP; r++)
            if (colok[r]) *((gptr)ep(st, 0, L * r) + uq) = rv[r];
        if (full) {
#pragma unroll
            for (int t = 0; t < MP; t++)
                if (L * t + q < rp) __builtin_nontemporal_store(rv[RP + t], (gptr)ep(es, a.L.es_innov, L * t) + uq);
        }
        // a non-finite result is stored as it is (helper.go:142-172 has no guard) and flagged
        if (chk != chk && q == 0) atomicOr(a.status + fi, (unsigned)KB_ST_NONFINITE);
    }
}

template <typename T, int N, int NM, int L>
#ifndef KB_SRIF_SPLIT_WAVES
#define KB_SRIF_SPLIT_WAVES ((N <= 8 && NM <= 4) ? 3 : 2)   // (a number: diagnostic builds)
#endif
__global__ void __launch_bounds__(64, KB_SRIF_SPLIT_WAVES) srif_split_kernel(const StepArgs a) {
    __shared__ __attribute__((aligned(16))) T lds[srif_split_lds_elems<(N + L - 1) / L * L, NM, L>()];
    // (a part that is HALF a 128-byte line -- eight lanes in fp64, four in fp32 -- shares every line with its neighbour: the two run on the same XCD,
    // kb_vanilla_split.h split_part_of_block; round 6: the fp32 kernels read 1.96x their bytes until this applied to them as well)
    constexpr int PAIRING = (64 / L) * (int)sizeof(T) < 128 ? 8 : 4;
    srif_split_part<T, N, NM, L>(a, split_part_of_block<PAIRING>(blockIdx.x, gridDim.x), lds);
}

// fp64: four lanes per filter up to 12 states, eight beyond (LDS: (NS^2 + NS) elements per filter; registers: NS / L rows of 2 NS values).
// fp32: four lanes throughout (half the registers and half the LDS per value).
template <typename T, int N, int NM>
static void srif_split_launch(const Batch &b, const StepArgs &a) {
    constexpr int L = (N <= 12 || sizeof(T) == 4) ? 4 : 8;   // (fp32 at 13..16 states, A/B: eight lanes 307 us at 13/2 and 376 at 16/8 against 250 / 311 on four)
    KB_LAUNCH((srif_split_kernel<T, N, NM, L>), dim3((unsigned)(a.ntiles * L)), dim3(64), 0, b.stream, a);
}
// one translation unit per group of state dimensions (kb_srif_split_*.hip): p <= 4, p <= 6 and p <= 8 instantiations of each
#define KB_SRIF_SPLIT_TU(N_)                                                                                              \
    void launch_srif_split_n##N_(const Batch &b, const StepArgs &a) {                                                     \
        if (a.p <= 4 || a.predict) srif_split_launch<double, N_, 4>(b, a);   /* (Predict() never sees the measurement) */ \
        else if (a.p <= 6) srif_split_launch<double, N_, 6>(b, a);                                                        \
        else srif_split_launch<double, N_, 8>(b, a);                                                                      \
    }
// n < 6: no p <= 6 instantiation (p = 5, 6 beside fewer than six states is no shape anybody times: library size)
#define KB_SRIF_SPLIT_TU_SMALL(N_)                                                                                        \
    void launch_srif_split_n##N_(const Batch &b, const StepArgs &a) {                                                     \
        if (a.p <= 4 || a.predict) srif_split_launch<double, N_, 4>(b, a);                                                \
        else srif_split_launch<double, N_, 8>(b, a);                                                                      \
    }
// fp32 (kb_srif_split_f32*.hip): the shapes the two-lane fp32 kernels do not serve -- odd n, n < 6, and Predict() / p = 7, 8 at 14, 16 states
#define KB_SRIF_SPLIT_TU_F32(N_)                                                                                          \
    void launch_srif_split_f32_n##N_(const Batch &b, const StepArgs &a) {                                                 \
        if (a.p <= 4 || a.predict) srif_split_launch<float, N_, 4>(b, a);                                                 \
        else srif_split_launch<float, N_, 8>(b, a);   /* (no p <= 6 instantiation in fp32: library size) */               \
    }
#undef KB_SB

}  // namespace kb
