// kb_squareroot_split.h -- SquareRoot.Update (squareroot.go:129-274) for 6 < n <= 16 with ONE FILTER SPLIT OVER L LANES.
//
// The one-filter-per-lane kernel (kb_squareroot_reg.hip) holds the 2n x n and (n + p) x (n + p) Householder panels in one lane's
// registers: 9 x 9 at 6 / 3, but 24 x 12 + 18 x 18 = 612 values at 12 / 6.  Here, as in kb_vanilla_split.h, a wave owns 64 / L
// filters (lane = q (64 / L) + f: 128-byte segments at L = 4), and the panels are distributed BY COLUMNS:
//
//   columns        lane q owns columns q, q + L, ... of C = [S^T F^T ; sqrtQ^T] and of Delta's state block, and measurement columns
//                  q, q + L, ... of Delta.  Column j of S^T F^T is row j of F S: (own row of F) x (all of S), S broadcast through LDS
//                  (packed, as P in the Vanilla kernel); column j of sqrtQ^T is row j of chol(Q): the lane's own rows.
//   Householder    (Dgeqr2 / Dlarfg as kb_static.h sqr_r states them) the reflector of column k is formed by the ONE lane that owns
//                  the column -- norm, beta, u0, 1 / (beta u0): no sum over lanes -- and handed to the other lanes of the filter
//                  through LDS (n + 2 values per step); every lane then applies it to its own columns right of k: the dot products
//                  u . C[:, c] are local as well.  Cyclic ownership keeps the shrinking set of active columns spread over the lanes.
//   results        Uc (= S-, the quirk of squareroot.go:185) goes to LDS once (packed) for Delta's S-^T H^T and S-^T blocks; of the
//                  second factor, W^T and S+^T are the top / bottom parts of the lane's own state columns (so K = W Syy^-1 and the
//                  store of S+ need nothing from other lanes); Syy (p x p) is gathered through LDS and inverted by every lane.
//
// Same operations in the same order as the register kernel per panel entry (the sums run over the rows of a column in ascending
// order, structural zeros skipped as ActC / ActD do); the padded family (GEN) relies on the same facts: a zero column makes no
// reflection, chol(R) is padded with an identity block.
#pragma once
#include "kb_vanilla_split.h"

namespace kb {

#define KB_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef KB_SQSPLIT_ACC
#define KB_SQSPLIT_ACC 1   // (4: the column norms in four partial sums -- A/B, profiles/NOTES.md round 5)
#endif

template <int NS, int NM>
constexpr int sqsplit_lds_elems() { return (tri(NS) > NS * NM ? tri(NS) : NS * NM) + 2 * NS + 2 + NS + NM + NM * NM; }   // S / Uc packed | the reflector of one column step (u0, f, then <= 2 n rows; n + p <= 2 n)

// Dlarfg for one column (kb_static.h sqr_r): given alpha = a[k][k] and the squared norm of the active entries below it, the
// unnormalised reflector H = I + f u u^T, u = (u0, x); returns the new diagonal entry
template <typename T>
__device__ __forceinline__ T reflector(T alpha, T xnorm2, bool more_rows, T &u0, T &f) {
    const bool refl = more_rows && (xnorm2 != T(0));
    const T beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
    u0 = alpha - beta;
#ifdef KB_SQSPLIT_IEEE_DIV
    f = refl ? T(1) / (beta * u0) : T(0);
#else
    f = refl ? recip(beta * u0) : T(0);   // kb_device.h: within an ulp of the quotient, a third of its instructions (one per Householder step)
#endif
    return refl ? beta : alpha;
}

// GEN: run-time dimensions n <= NS, p <= NM, m <= NC (zero padding); RT (GEN only): FULL and the Noise come from the launch arguments
// as well -- one instantiation per NS for everything, at one wave per SIMD; RT = false: Noiseless, FULL as given (kb_vanilla_split.h)
template <typename T, int NS, int NM, int NC, int L, bool GEN, bool FULLT, bool RT = GEN, bool NOISET = false>
__device__ __forceinline__ void squareroot_split_part(const StepArgs &a, const int64_t gw, T *lds) {
    static_assert(NS % L == 0, "columns are dealt out cyclically");
    constexpr int FPW = 64 / L, RP = NS / L, PC = (NM + L - 1) / L, TR = tri(NS), TM = tri(NM), DD = NS + NM;
    constexpr int BOFF = TR > NS * NM ? TR : NS * NM;   // LDS: [0, BOFF) S, later Uc, later W (n x p, parked as its rows become final);
    constexpr int XOFF = BOFF + 2 * NS + 2;   // [BOFF, XOFF) the reflector of the current step; then x- (n) and H x- (p), parked until the end;
    constexpr int SYOFF = XOFF + NS + NM;    // then Syy^T (p x p), written entry by entry as the measurement columns become final
    constexpr bool XPARK = NS <= NM * NM;
    typedef __attribute__((address_space(1))) T *gptr;
    const int rn = GEN ? a.n : NS, rp = GEN ? a.p : NM, rm = GEN ? (a.need_ctrl ? a.m : 0) : NC;
    const bool full = RT ? (a.flags & KB_FLAG_FULL_ESTIMATE) != 0 : FULLT;
    const bool awgn = NOISET || (RT && a.noise_kind == KB_NOISE_AWGN);   // (NOISET: AWGN at compile time, kb_vanilla_split.h)
    const unsigned lane = threadIdx.x;
    const int q = (int)((lane / FPW) & (L - 1)), f = (int)(lane & (FPW - 1));
    const int64_t tile = gw / L;
    const int slot = (int)(gw % L) * FPW + f;
    if (tile * KB_TILE + (gw % L) * FPW >= a.N) return;
    const bool active = tile * KB_TILE + slot < a.N;

    T *const st = (T *)a.state + tile * ((int64_t)KB_TILE * (rn + tri(rn)));
    const T *const mo = (const T *)a.model + tile * a.mo_ts;
    const unsigned us = (unsigned)slot;
    const unsigned um = a.mo_ts ? (unsigned)slot : 0u;
    const unsigned uq = us + (unsigned)(q * KB_TILE);
    const unsigned uf = um + (unsigned)(q * rn * KB_TILE);   // row q of an n-column matrix of the model block (F, and H's row q)
    // LDS layout: TWO consecutive elements per lane (16 bytes), element e at lp[PX(e)]: the contiguous runs this kernel reads -- a reflector, a
    // row of the packed factor -- go as ds_read_b128, which the LDS array serves at twice the bytes per clock of the ds_read2_b64 pairs the
    // compiler forms from 8-byte neighbours (kb_vanilla_split.h PAIRED; NOTES.md: the array is busy ~50 % of the kernel).
#ifdef KB_SQSPLIT_UNPAIRED
    constexpr bool PAIRED = false;
#else
    constexpr bool PAIRED = true;
#endif
    T *const lp = PAIRED ? lds + 2 * f : lds + f;
    auto PX = [](int e) constexpr -> int { return PAIRED ? (e >> 1) * (2 * FPW) + (e & 1) : e * FPW; };
    // element (B + c), B a per-lane element number and c a compile-time constant: one base for even c, one for odd c
    struct DynBase { T *e, *o; };
    auto dyn = [&](int B) -> DynBase {
        if (!PAIRED) return DynBase{lp + B * FPW, lp + B * FPW};
        const int pb = (B >> 1) * (2 * FPW) + (B & 1);
        return DynBase{lp + pb, lp + ((B & 1) ? ((B + 1) >> 1) * (2 * FPW) : pb + 1)};
    };
    auto at = [&](const DynBase &d, int c) -> T & { if (!PAIRED) return d.e[c * FPW]; return (c & 1) ? d.o[PX(c - 1)] : d.e[PX(c)]; };
    const DynBase dq = dyn(q);
    auto ep = [&](const T *ubase, int rt, int c) -> gptr { return (gptr)anchored(ubase, rt, c); };
    // (model streams: non-temporal where a lane group reads whole 128-byte segments (L <= 4); with eight lanes per filter a group reads HALF
    // a line and the part next door the other half a little later -- the streaming hint lets the line leave the L2 in between and it comes
    // from memory twice (kb_srif_split.h: 1.36x the packed reads with the hint, 1.04x without), so there the default policy)
    auto ldg = [&](const T *ubase, int rt, int c, unsigned off) { if constexpr (L == 8) return *(ep(ubase, rt, c) + off); else return __builtin_nontemporal_load(ep(ubase, rt, c) + off); };
    bool colok[RP], colany[RP];   // own column / row j_r = q + L r is a real one; some lane's is (wave-uniform)
#pragma unroll
    for (int r = 0; r < RP; r++) { colok[r] = !GEN || q + L * r < rn; colany[r] = !GEN || L * r < rn; }
    unsigned utri[RP];   // 64 tri(j_r): row j_r of a packed lower-triangular matrix (chol Q, S) starts there
#pragma unroll
    for (int r = 0; r < RP; r++) utri[r] = (unsigned)(((q + L * r) * (q + L * r + 1) / 2) * KB_TILE);

    // ---- phase 0: F (own rows), x, S (a packed share per lane, handed to LDS) ----------------------------------------------------
    T Fo[RP][NS], x[NS];
    {
        constexpr int KP = (TR + L - 1) / L;
        T Sp[KP];
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int l = 0; l < NS; l++) {
                const T v = (colany[r] && l < rn) ? ldg(mo, a.L.mo_F + (GEN ? L * r * rn : 0), (GEN ? 0 : L * r * NS) + l, colok[r] ? uf : um) : T(0);
                Fo[r][l] = colok[r] ? v : T(0);
            }
        auto load_state = [&](auto NT) {
#pragma unroll
            for (int k = 0; k < KP; k++) {
                const bool okp = L * k + q < tri(rn);
                const gptr pe = ep(st, rn, L * k) + (okp ? uq : us);
                const T v = L * k < tri(rn) ? (decltype(NT)::value ? __builtin_nontemporal_load(pe) : *pe) : T(0);
                Sp[k] = okp ? v : T(0);
            }
#pragma unroll
            for (int l = 0; l < NS; l++) {
                const gptr pe = ep(st, 0, l) + us;
                x[l] = l < rn ? (decltype(NT)::value ? __builtin_nontemporal_load(pe) : *pe) : T(0);
            }
        };
        KB_WITH_STATE_POLICY(a, load_state);
        KB_SB();
#pragma unroll
        for (int k = 0; k < KP; k++)
            if (L * k + L - 1 < TR || L * k + q < TR) at(dq, L * k) = Sp[k];
        if constexpr (XPARK) {   // FULL: x_prev waits in Syy's slots (free until the second factorisation) for yhat = H x_prev
            if (full) {
#pragma unroll
                for (int l = 0; l < NS; l++) lp[PX(SYOFF + l)] = x[l];   // (the L lanes of a filter write the same values)
            }
        }
    }
    wave_lds_fence();
    KB_SB();

    // ---- phase 1: x- = F x [+ G u]; the top block of C, own columns: C[i][j] = sum_{l >= i} S[l][i] F[j][l] (squareroot.go:155-175),
    // one row of S per chunk (row l of S is packed contiguously: S[l][i] at tri(l) + i) ------------------------------------------------
    T xm[RP], Ct[RP][NS], Cb[RP][NS];
#pragma unroll
    for (int r = 0; r < RP; r++) {
        T s = T(0);
#pragma unroll
        for (int l = 0; l < NS; l++) s += Fo[r][l] * x[l];
        xm[r] = s;
        pin(xm[r]);
#pragma unroll
        for (int i = 0; i < NS; i++) Ct[r][i] = T(0);
    }
    if constexpr (NC > 0) {
        if (rm > 0) {
            const T *up = (const T *)a.u + tile * a.u_ts;
            T u[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) u[c] = (active && c < rm) ? ldnt_at(&(up + (int64_t)c * a.u_es)[us]) : T(0);
#pragma unroll
            for (int r = 0; r < RP; r++) {
                T s = T(0);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const T g = (colany[r] && c < rm) ? ldg(mo, a.L.mo_G + L * r * rm, c, colok[r] ? um + (unsigned)(q * rm * KB_TILE) : um) : T(0);
                    s += (colok[r] ? g : T(0)) * u[c];
                }
                xm[r] = xm[r] + s;
            }
        }
    }
    {
        T row[2][NS];
        auto fetch = [&](int l, int b) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NS; i++)
                if (i <= l) row[b][i] = lp[PX(l * (l + 1) / 2 + i)];
        };
        fetch(0, 0);
        // (sfor, kb_device.h: the bounds of the triangular inner loops are compile-time constants from the start; as `#pragma unroll`
        // loops inside a loop they are unrolled at run time with remainder loops BEFORE the outer loop is, and the panels land in scratch)
        sfor<0, NS>([&](auto LL) __attribute__((always_inline)) {
            constexpr int l = LL;
            if constexpr (l + 1 < NS) fetch(l + 1, (l + 1) & 1);
            KB_SB();
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int i = 0; i <= l; i++) { Ct[r][i] += row[l & 1][i] * Fo[r][l]; pin(Ct[r][i]); }
            KB_SB();
        });
    }
    // the bottom block: column j of sqrtQ^T = row j of chol(Q) (the lane's own rows, packed contiguously), zero below the diagonal
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T v = T(0);
            if (colany[r] && i < rn && i < L * r + L) v = ldg(mo, a.L.mo_LQ, i, (colok[r] && i <= q + L * r) ? um + utri[r] : um);
            Cb[r][i] = (colok[r] && i <= q + L * r) ? v : T(0);
        }
    KB_SB();

    // ---- Householder on C (2n x n; Dgeqr2): column k's owner forms the reflector, everybody applies it to the own columns c > k --------
    wave_lds_fence();   // (the S reads are done; the reflector buffer is a different region, but the compiler does not know the lanes)
    sfor<0, NS>([&](auto KK) __attribute__((always_inline)) {
        constexpr int k = KK, rk = k / L, qk = k % L;
        {   // the owner's part, computed by every lane on its local column rk; only lane group qk publishes it
            T xn2;
            if constexpr (KB_SQSPLIT_ACC > 1) {   // (partial sums: the dependent chain of n + p FMAs in front of the square root is the step's longest)
                T x4[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int i = k + 1; i < NS; i++) x4[(i - (k + 1)) & 3] += Ct[rk][i] * Ct[rk][i];
                xn2 = (x4[0] + x4[1]) + (x4[2] + x4[3]);
            } else {
                xn2 = T(0);
#pragma unroll
                for (int i = k + 1; i < NS; i++) xn2 += Ct[rk][i] * Ct[rk][i];
            }
            if constexpr (KB_SQSPLIT_ACC > 1) {
                T y2[2] = {T(0), T(0)};
#pragma unroll
                for (int i = 0; i <= k; i++) y2[i & 1] += Cb[rk][i] * Cb[rk][i];
                xn2 += y2[0] + y2[1];
            } else {
#pragma unroll
                for (int i = 0; i <= k; i++) xn2 += Cb[rk][i] * Cb[rk][i];
            }
            T u0, fr;
            const T dg = reflector<T>(Ct[rk][k], xn2, true, u0, fr);
            if (q == qk) {
                lp[PX(BOFF + 0)] = u0;
                lp[PX(BOFF + 1)] = fr;
#pragma unroll
                for (int i = k + 1; i < NS; i++) lp[PX(BOFF + 2 + i)] = Ct[rk][i];
#pragma unroll
                for (int i = 0; i <= k; i++) lp[PX(BOFF + 2 + NS + i)] = Cb[rk][i];   // (the bottom rows sit behind the n top slots)
                Ct[rk][k] = dg;
            }
        }
        wave_lds_fence();
        {
            const T u0 = lp[PX(BOFF + 0)], fr = lp[PX(BOFF + 1)];
            T ut[NS], ub[NS];
#pragma unroll
            for (int i = k + 1; i < NS; i++) ut[i] = lp[PX(BOFF + 2 + i)];
#pragma unroll
            for (int i = 0; i <= k; i++) ub[i] = lp[PX(BOFF + 2 + NS + i)];
#pragma unroll
            for (int r = rk; r < RP; r++) {
                const bool upd = r > rk || q > qk;   // own column q + L r lies right of k
                T s = u0 * Ct[r][k];
#pragma unroll
                for (int i = k + 1; i < NS; i++) s += ut[i] * Ct[r][i];
#pragma unroll
                for (int i = 0; i <= k; i++) s += ub[i] * Cb[r][i];
                const T fs = upd ? fr * s : T(0);
                Ct[r][k] += fs * u0;
#pragma unroll
                for (int i = k + 1; i < NS; i++) Ct[r][i] += fs * ut[i];
#pragma unroll
                for (int i = 0; i <= k; i++) Cb[r][i] += fs * ub[i];
            }
#pragma unroll
            for (int r = rk; r < RP; r++)
#pragma unroll
                for (int i = 0; i < NS; i++) { pin(Ct[r][i]); pin(Cb[r][i]); }
        }
        wave_lds_fence();
        KB_SB();
    });
    // Uc[i][j] = Ct[r][i], i <= j = j_r: S- := Uc (QUIRK squareroot.go:185).  It goes to LDS packed by columns (tri(j) + i): the own
    // columns are contiguous.  FULL: the Estimate's predicted factor leaves at once (as in kb_squareroot_reg.hip).
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int i = 0; i < L * r + L; i++)
            if (i <= q + L * r) at(dyn((q + L * r) * (q + L * r + 1) / 2), i) = Ct[r][i];
    if (full && active) {
        T *const es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems);
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int i = 0; i < L * r + L; i++)
                if (colok[r] && i <= q + L * r) __builtin_nontemporal_store(Ct[r][i], ep(es, a.L.es_ppred, i) + (us + utri[r]));
    }
    wave_lds_fence();
    KB_SB();

    // ---- Delta = [[sqrtR^T, 0], [S-^T H^T, S-^T]] (squareroot.go:190-216), by columns ------------------------------------------------
    // measurement columns c = q + L r2: top = row c of chol(R) (own row), bottom (S-^T H^T)[i][c] = sum_{l <= i} Uc[l][i] H[c][l]
    T Dm[PC][DD], Ds[RP][DD];
    bool mcol[PC];   // own measurement column exists (c < NM); real (c < rp) or identity padding
    [[maybe_unused]] T hxp[PC];   // FULL: (H x_prev)[c] for the own measurement rows
    {
        T Hrow[PC][NS];
#pragma unroll
        for (int r2 = 0; r2 < PC; r2++) {
            const int cbase = L * r2;
            mcol[r2] = cbase + L - 1 < NM || q + cbase < NM;
            const bool real = mcol[r2] && q + cbase < rp;
            const unsigned urow = real ? um + (unsigned)((q + cbase) * rn * KB_TILE) : um;
#pragma unroll
            for (int l = 0; l < NS; l++) {
                const T v = (cbase < rp && l < rn) ? ldg(mo, a.L.mo_H, l, urow) : T(0);
                Hrow[r2][l] = real ? v : T(0);
            }
            const unsigned utr = real ? um + (unsigned)(((q + cbase) * (q + cbase + 1) / 2) * KB_TILE) : um;
#pragma unroll
            for (int rr = 0; rr < NM; rr++) {
                T v = T(0);
                if (cbase < rp && rr < cbase + L && rr < rp) v = ldg(mo, a.L.mo_LR, rr, utr);
                Dm[r2][rr] = real ? (rr <= q + cbase ? v : T(0)) : ((mcol[r2] && rr == q + cbase) ? T(1) : T(0));
            }
#pragma unroll
            for (int i = 0; i < NS; i++) Dm[r2][NM + i] = T(0);
        }
        if (full) {   // squareroot.go:237-239 yhat = H x_prev: x_prev from its LDS slots (or read a second time: a cache hit), not kept through the first factorisation
#pragma unroll
            for (int r2 = 0; r2 < PC; r2++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) {
                    if constexpr (XPARK) s += Hrow[r2][l] * lp[PX(SYOFF + l)];
                    else s += Hrow[r2][l] * ((l < rn) ? *(ep(st, 0, l) + us) : T(0));
                }
                hxp[r2] = s;
                pin(hxp[r2]);
            }
            KB_SB();
        }
        // bottom rows, one column of Uc per chunk (column i of Uc is packed contiguously: Uc[l][i] at tri(i) + l)
        T col[2][NS];
        auto fetch = [&](int i, int b) __attribute__((always_inline)) {
#pragma unroll
            for (int l = 0; l < NS; l++)
                if (l <= i) col[b][l] = lp[PX(i * (i + 1) / 2 + l)];
        };
        fetch(0, 0);
        sfor<0, NS>([&](auto II) __attribute__((always_inline)) {
            constexpr int i = II;
            if constexpr (i + 1 < NS) fetch(i + 1, (i + 1) & 1);
            KB_SB();
#pragma unroll
            for (int r2 = 0; r2 < PC; r2++) {
                T s = T(0);
#pragma unroll
                for (int l = 0; l <= i; l++) s += col[i & 1][l] * Hrow[r2][l];
                Dm[r2][NM + i] = s;
                pin(Dm[r2][NM + i]);
            }
            KB_SB();
        });
        // H x- for the innovation (squareroot.go:255-262): x- is gathered through LDS (its rows are spread over the lanes), the own
        // measurement rows are formed here, and gathered below
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < RP; r++) at(dq, XOFF + L * r) = xm[r];
        wave_lds_fence();
        T hx[PC];
#pragma unroll
        for (int r2 = 0; r2 < PC; r2++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) s += Hrow[r2][l] * lp[PX(XOFF + l)];
            hx[r2] = s;
        }
        wave_lds_fence();
#pragma unroll
        for (int r2 = 0; r2 < PC; r2++)
            if (mcol[r2]) at(dq, XOFF + NS + L * r2) = hx[r2];
    }
    // state columns p + j, j = j_r: top zero, bottom S-^T[i][j] = Uc[j][i] for i >= j (element tri(i) + j of the packed Uc)
#pragma unroll
    for (int r = 0; r < RP; r++) {
#pragma unroll
        for (int rr = 0; rr < NM; rr++) Ds[r][rr] = T(0);
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T v = T(0);
            if (i >= L * r) v = at(dq, i * (i + 1) / 2 + L * r);
            Ds[r][NM + i] = (i >= q + L * r) ? v : T(0);
        }
    }
    wave_lds_fence();
    KB_SB();

    // ---- Householder on Delta ((n + p) x (n + p); ActD: below the diagonal the first p columns are zero in their first p rows) -------
    // (a) the measurement columns k < p: active rows below the diagonal are the n bottom rows; every state column is updated
    sfor<0, NM>([&](auto KK) __attribute__((always_inline)) {
        constexpr int k = KK, rk = k / L, qk = k % L;
        {
            T xn2;
            if constexpr (KB_SQSPLIT_ACC > 1) {   // (partial sums: the dependent chain of n + p FMAs in front of the square root is the step's longest)
                T x4[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int i = NM; i < DD; i++) x4[(i - (NM)) & 3] += Dm[rk][i] * Dm[rk][i];
                xn2 = (x4[0] + x4[1]) + (x4[2] + x4[3]);
            } else {
                xn2 = T(0);
#pragma unroll
                for (int i = NM; i < DD; i++) xn2 += Dm[rk][i] * Dm[rk][i];
            }
            T u0, fr;
            const T dg = reflector<T>(Dm[rk][k], xn2, true, u0, fr);
            if (q == qk) {
                lp[PX(BOFF + 0)] = u0;
                lp[PX(BOFF + 1)] = fr;
#pragma unroll
                for (int i = NM; i < DD; i++) lp[PX(BOFF + 2 + i)] = Dm[rk][i];
                Dm[rk][k] = dg;
            }
        }
        wave_lds_fence();
        {
            const T u0 = lp[PX(BOFF + 0)], fr = lp[PX(BOFF + 1)];
            T u[DD];
#pragma unroll
            for (int i = NM; i < DD; i++) u[i] = lp[PX(BOFF + 2 + i)];
#pragma unroll
            for (int r2 = rk; r2 < PC; r2++) {
                const bool upd = r2 > rk || q > qk;
                T s = u0 * Dm[r2][k];
#pragma unroll
                for (int i = NM; i < DD; i++) s += u[i] * Dm[r2][i];
                const T fs = upd ? fr * s : T(0);
                Dm[r2][k] += fs * u0;
#pragma unroll
                for (int i = NM; i < DD; i++) { Dm[r2][i] += fs * u[i]; pin(Dm[r2][i]); }
                pin(Dm[r2][k]);
            }
#pragma unroll
            for (int r = 0; r < RP; r++) {
                T s = u0 * Ds[r][k];
#pragma unroll
                for (int i = NM; i < DD; i++) s += u[i] * Ds[r][i];
                const T fs = fr * s;
                Ds[r][k] += fs * u0;
#pragma unroll
                for (int i = NM; i < DD; i++) { Ds[r][i] += fs * u[i]; pin(Ds[r][i]); }
                // row k of the state columns is final (W[j_r][k] = UD[k][p + j_r]): it waits in LDS for the gain, not in a register
                (lp + q * NM * FPW)[PX((L * r) * NM + k)] = Ds[r][k];
            }
            // ... and so is row k of the measurement columns c >= k (Syy[c][k] = UD[k][c]): slot NM c + k of the Syy region
#pragma unroll
            for (int r2 = rk; r2 < PC; r2++)
                if (mcol[r2] && (r2 > rk || q >= qk)) (lp + q * NM * FPW)[PX(SYOFF + NM * (L * r2) + k)] = Dm[r2][k];
        }
        wave_lds_fence();
        KB_SB();
    });
    // ---- Syy = UD[:p,:p]^T gathered through LDS, K = W Syy^-1 (squareroot.go:225-252; the inverse's error is never looked at).  W^T is
    // the top part of the own state columns: W[j_r][k2] = UD[k2][p + j_r] = Ds[r][k2].  Both are FINAL once the measurement columns are
    // done, so the gain and x+ are formed here, before the state columns are factorised: W would otherwise wait in registers through (b)
    wave_lds_fence();
    T K[RP][NM];
    {
        // (the inverse column by column -- lu_factor_any / lu_inverse_column, kb_vanilla_split.h -- so that it never exists as a whole
        // next to its factors: the bottom parts of the state columns, 36 values, are waiting in registers for phase (b))
        T Syy[NM * NM];
#pragma unroll
        for (int i = 0; i < NM; i++)
#pragma unroll
            for (int j = 0; j < NM; j++) Syy[i * NM + j] = (j <= i) ? lp[PX(SYOFF + NM * i + j)] : T(0);   // Syy[i][j] = UD[j][i]
        unsigned swaps;
        T anorm;
        (void)lu_factor_any<T, NM>(Syy, swaps, anorm, rp);
        T Wr[RP][NM];
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int k2 = 0; k2 < NM; k2++) Wr[r][k2] = (lp + q * NM * FPW)[PX((L * r) * NM + k2)];   // W[j_r][k2], parked above
        sfor<0, NM>([&](auto CC) __attribute__((always_inline)) {
            constexpr int c = CC;
            T v[NM];
            lu_inverse_column<T, NM, c>(Syy, swaps, v);
#pragma unroll
            for (int r = 0; r < RP; r++) {
                T s = T(0);
#pragma unroll
                for (int k2 = 0; k2 < NM; k2++) s += Wr[r][k2] * v[k2];
                K[r][c] = s;
                pin(K[r][c]);
            }
            KB_SB();
        });
    }
    // ---- squareroot.go:255-268 x+ = x- + K (y - H x-) [+ Process(k)] -----------------------------------------------------------------
    T innov[NM], xn[RP];
    {
        const T *yp = (const T *)a.y + tile * a.y_ts;
#pragma unroll
        for (int c = 0; c < NM; c++) innov[c] = ((active && c < rp) ? ldnt_at(&(yp + (int64_t)c * a.y_es)[us]) : T(0)) - lp[PX(XOFF + NS + c)];
    }
#pragma unroll
    for (int r = 0; r < RP; r++) {
        T s = T(0);
#pragma unroll
        for (int c = 0; c < NM; c++) s += K[r][c] * innov[c];
        xn[r] = at(dq, XOFF + L * r) + s;   // x-[j_r], parked in LDS until here
        pin(xn[r]);
    }
    if (full) {   // the Estimate's gain and innovation wait in LDS (W's and Syy's slots: both consumed) for the end of the step
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int c = 0; c < NM; c++) (lp + q * NM * FPW)[PX((L * r) * NM + c)] = K[r][c];
        if (q == 0) {
#pragma unroll
            for (int c = 0; c < NM; c++) lp[PX(SYOFF + c)] = innov[c];
        }
    }
    KB_SB();
    // (b) the state columns k = p + kk: all rows below the diagonal are active; the last column makes no reflection (Dgeqr2: M - i > 1)
    sfor<0, NS>([&](auto KK) __attribute__((always_inline)) {
        constexpr int kk = KK, k = NM + kk, rk = kk / L, qk = kk % L;
        {
            T xn2;
            if constexpr (KB_SQSPLIT_ACC > 1) {   // (partial sums: the dependent chain of n + p FMAs in front of the square root is the step's longest)
                T x4[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int i = k + 1; i < DD; i++) x4[(i - (k + 1)) & 3] += Ds[rk][i] * Ds[rk][i];
                xn2 = (x4[0] + x4[1]) + (x4[2] + x4[3]);
            } else {
                xn2 = T(0);
#pragma unroll
                for (int i = k + 1; i < DD; i++) xn2 += Ds[rk][i] * Ds[rk][i];
            }
            T u0, fr;
            const T dg = reflector<T>(Ds[rk][k], xn2, k + 1 < DD, u0, fr);
            if (q == qk) {
                lp[PX(BOFF + 0)] = u0;
                lp[PX(BOFF + 1)] = fr;
#pragma unroll
                for (int i = k + 1; i < DD; i++) lp[PX(BOFF + 2 + i)] = Ds[rk][i];
                Ds[rk][k] = dg;
            }
        }
        wave_lds_fence();
        {
            const T u0 = lp[PX(BOFF + 0)], fr = lp[PX(BOFF + 1)];
            T u[DD];
#pragma unroll
            for (int i = k + 1; i < DD; i++) u[i] = lp[PX(BOFF + 2 + i)];
#pragma unroll
            for (int r = rk; r < RP; r++) {
                const bool upd = r > rk || q > qk;
                T s = u0 * Ds[r][k];
#pragma unroll
                for (int i = k + 1; i < DD; i++) s += u[i] * Ds[r][i];
                const T fs = upd ? fr * s : T(0);
                Ds[r][k] += fs * u0;
#pragma unroll
                for (int i = k + 1; i < DD; i++) { Ds[r][i] += fs * u[i]; pin(Ds[r][i]); }
                pin(Ds[r][k]);
            }
        }
        wave_lds_fence();
        KB_SB();
    });

    [[maybe_unused]] T vown[PC];
#pragma unroll
    for (int r2 = 0; r2 < PC; r2++) vown[r2] = T(0);
    if constexpr (RT || NOISET) {
        if (awgn) {   // noise.go:109-164: Process(k) into x+ (which = 2), Measurement(k) into yhat (which = 1, FULL only); the factors are read again
            const uint64_t gfi = (uint64_t)(a.first_filter + tile * KB_TILE) + (unsigned)slot;
            const uint32_t stepno = (uint32_t)a.step0 - (active ? a.lag[tile * KB_TILE + slot] : 0u);   // kf.step of this filter
            // the L lanes of a filter share its Box-Muller blocks and gather the vector through LDS (kb_vanilla_split.h draw_coop; the
            // reflector buffer at BOFF is free by now: 2 n + 2 >= n slots)
            auto draw_coop = [&](auto NVC, uint32_t which, auto &zv) __attribute__((always_inline)) {
                constexpr int NV = decltype(NVC)::value, NB = (NV + 1) / 2, IT = (NB + L - 1) / L;
                wave_lds_fence();
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    const int blk = q + L * it;
                    uint32_t rr[4];
                    Philox::gen(a.seed, gfi, stepno, ((uint32_t)(a.epoch * 4 + which) << 8) | (uint32_t)blk, rr);
                    double z0, z1;
                    box_muller(rr, z0, z1);
                    if (L * it + L - 1 < NB || blk < NB) {
                        lp[PX(BOFF + 2 * blk)] = (T)z0;
                        if (2 * (L * it + L - 1) + 1 < NV || 2 * blk + 1 < NV) lp[PX(BOFF + 2 * blk + 1)] = (T)z1;
                    }
                    KB_SB();
                }
                wave_lds_fence();
#pragma unroll
                for (int kk = 0; kk < NV; kk++) zv[kk] = lp[PX(BOFF + kk)];
                wave_lds_fence();
            };
            {
                T z[NS];
                draw_coop(std::integral_constant<int, NS>{}, 2u, z);
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    T sacc = T(0);
#pragma unroll
                    for (int i = 0; i < L * r + L; i++) {
                        const bool in = colok[r] && i <= q + L * r;
                        const T l = (colany[r] && i < rn) ? ldg(mo, a.L.mo_LQ, i, in ? um + utri[r] : um) : T(0);
                        sacc += (in ? l : T(0)) * z[i];
                    }
                    xn[r] += sacc;
                }
            }
            if (full) {
                T z1[NM];
                draw_coop(std::integral_constant<int, NM>{}, 1u, z1);
#pragma unroll
                for (int r2 = 0; r2 < PC; r2++) {
                    const int cbase = L * r2;
                    const bool real = mcol[r2] && q + cbase < rp;
                    const unsigned utr = real ? um + (unsigned)(((q + cbase) * (q + cbase + 1) / 2) * KB_TILE) : um;
                    T sacc = T(0);
#pragma unroll
                    for (int i = 0; i < NM; i++) {
                        const bool in = real && i <= q + cbase;
                        const T l = (cbase < rp && i < cbase + L && i < rp) ? ldg(mo, a.L.mo_LR, i, utr) : T(0);
                        sacc += (in ? l : T(0)) * z1[i];
                    }
                    vown[r2] = sacc;
                }
            }
        }
    }
    // ---- non-finite screen over the own entries of x+ and S+ (S+[i][j] = UD[p + j][p + i]: the own column's bottom part), all lanes
    unsigned err;
    {
        T chk = T(0);
#pragma unroll
        for (int r = 0; r < RP; r++) {
            chk += xn[r] * T(0);
#pragma unroll
            for (int j = 0; j < L * r + L; j++) chk += ((j <= q + L * r) ? Ds[r][NM + j] : T(0)) * T(0);
        }
        err = sum_lanes<L>((chk != chk) ? (unsigned)KB_ST_NONFINITE : 0u);
    }
    const bool ok = err == 0;
    T *const es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems);
    if (active && ok) {
        auto store_state = [&](auto NT) {
#pragma unroll
            for (int r = 0; r < RP; r++) {
                if (colok[r]) {
                    const gptr pe = ep(st, 0, L * r) + uq;
                    if constexpr (decltype(NT)::value) __builtin_nontemporal_store(xn[r], pe); else *pe = xn[r];
                }
#pragma unroll
                for (int j = 0; j < L * r + L; j++)
                    if (colok[r] && j <= q + L * r) {   // row i = j_r of S+ is packed contiguously at tri(i)
                        const gptr pe = ep(st, rn, j) + (us + utri[r]);
                        if constexpr (decltype(NT)::value) __builtin_nontemporal_store(Ds[r][NM + j], pe); else *pe = Ds[r][NM + j];
                    }
            }
        };
        KB_WITH_STATE_POLICY(a, store_state);
        if (full) {
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int c = 0; c < NM; c++)
                    if (colok[r] && c < rp) __builtin_nontemporal_store((lp + q * NM * FPW)[PX((L * r) * NM + c)], ep(es, a.L.es_gain + L * r * a.pmax, c) + (us + (unsigned)(q * a.pmax * KB_TILE)));
            if (q == 0) {
#pragma unroll
                for (int c = 0; c < NM; c++)
                    if (c < rp) __builtin_nontemporal_store(lp[PX(SYOFF + c)], ep(es, a.L.es_innov, c) + us);
            }
        }
    }
    if (full && active) {   // yhat leaves whether or not the step is applied (as in kb_squareroot_reg.hip): the owner of row c stores it
#pragma unroll
        for (int r2 = 0; r2 < PC; r2++)
            if (mcol[r2] && q + L * r2 < rp) __builtin_nontemporal_store(hxp[r2] + vown[r2], ep(es, a.L.es_yhat, L * r2) + uq);
    }
    const unsigned lane_end = late_lane();
    if (active && !ok && ((lane_end / FPW) & (L - 1)) == 0)
        atomicOr(a.status + tile * KB_TILE + (int64_t)((gw % L) * FPW + (lane_end & (FPW - 1))), (unsigned)KB_ST_NONFINITE);
    (void)TM;
}

template <typename T, int NS, int NM, int NC, int L, bool GEN, bool FULLT, bool RT = GEN, bool NOISET = false>
__global__ void __launch_bounds__(64, ((RT || NM > 6) && L == 4) ? 1 : 2) squareroot_split_kernel(const StepArgs a) {
    __shared__ __attribute__((aligned(16))) T lds[(sqsplit_lds_elems<NS, NM>() + 1) / 2 * 2 * (64 / L)];   // (whole pairs)
    squareroot_split_part<T, NS, NM, NC, L, GEN, FULLT, RT, NOISET>(a, split_part_of_block<L>(blockIdx.x, gridDim.x), lds);   // (kb_vanilla_split.h: XCD-aware for L = 8)
}
#undef KB_SB

}  // namespace kb
