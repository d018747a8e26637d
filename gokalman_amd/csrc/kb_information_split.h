// kb_information_split.h -- Information.Update (information.go:153-227) for 6 < n <= 16 with ONE FILTER SPLIT OVER L LANES.
// Mapping as kb_vanilla_split.h: a wave owns 64 / L filters, lane = q (64 / L) + f, lane q owns rows q, q + L, ...
//
//   M = F^-T I F^-1   (:163-165) two products "own rows x whole matrix": F^-1, then T1 = I F^-1, broadcast through LDS (one n x n
//                     region, reused); the own COLUMNS of F^-1 (= rows of F^-T) are read back from that region.
//   Z = -M (M + Q^-1)^-1  (:169-174; the reference ignores the inverse's error) as kb_information_reg.hip does it: the pivoted LU
//                     solve (M + Q^-1)^T Y = M^T, Z = -Y^T.  Column c of (M + Q^-1)^T is ROW c of B = M + Q^-1 and column c of the
//                     right-hand side is row c of M: both are the lane's own rows, so the pivot search of step k is local to the
//                     owner of row k, which hands the pivot index and the multipliers to the other lanes through LDS; the row
//                     exchange (entries k and piv of every own row: a per-filter index, select chains, run only when some filter of
//                     the wave pivots) and the elimination are local again.  Back substitution needs all of U: row by row through LDS.
//   I- = M + Z M^T, i- = (1 + Z)(F^-T i [+ M G u])  (:176-190) M broadcast through LDS; F^-T i gathered through LDS.
//   i+ = H^T R^-1 y + i-, I+ = I- + H^T R^-1 H  (:197-212) own columns of H in registers, H broadcast through LDS.
//   FULL (KB_FLAG_FULL_ESTIMATE): yhat = H State(prev) [+ Measurement(k)] (:192-194), State(prev) = inverse(I) i with the inverse
//                     mirrored from its upper triangle, or zeros when gonum's Inverse reports a Condition error (:284-288): the SAME
//                     distributed solve on I X = 1 at the top of the kernel, while every register and all of the LDS are free; the
//                     inverse goes to LDS whole, each lane reads the mirrored rows it owns.  I- leaves for the Estimate as it is formed.
// The sums run in the reference's order; LAPACK's pivot choice (first largest entry, one exchange per column).
#pragma once
#include "kb_vanilla_split.h"

namespace kb {

#define KB_SB() __builtin_amdgcn_sched_barrier(0)

template <int NS>
constexpr int infsplit_lds_elems() { return NS * NS + NS + 4; }   // one n x n operand | the hand-over of one elimination / substitution step

template <typename T, int NS, int NM, int NC, int L, bool GEN, bool FULLT = false>
__device__ __forceinline__ void information_split_part(const StepArgs &a, const int64_t gw, T *lds) {
    static_assert(NS % L == 0, "rows are dealt out cyclically");
    constexpr int FPW = 64 / L, RP = NS / L, TM = tri(NM), PC = (NM + L - 1) / L;
    constexpr int BOFF = NS * NS;   // LDS: [0, n^2) the broadcast operand of the phase; [n^2, n^2 + n + 4) hand-over buffer
    typedef __attribute__((address_space(1))) T *gptr;
    const int rn = GEN ? a.n : NS, rp = GEN ? a.p : NM, rm = GEN ? (a.need_ctrl ? a.m : 0) : NC;
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
    const unsigned umq = um + (unsigned)(q * KB_TILE);
    const unsigned uf = um + (unsigned)(q * rn * KB_TILE);
    // LDS layout: TWO consecutive elements per lane (16 bytes), element e at lp[PX(e)]: the contiguous runs this kernel reads -- the multipliers
    // of an elimination step, a row of an n x n operand -- go as ds_read_b128 at twice the array rate of the ds_read2_b64 pairs the compiler
    // forms from 8-byte neighbours (kb_vanilla_split.h PAIRED; NOTES.md).
#ifdef KB_INFSPLIT_UNPAIRED
    constexpr bool PAIRED = false;
#else
    constexpr bool PAIRED = true;
#endif
    T *const lp = PAIRED ? lds + 2 * f : lds + f;
    auto PX = [](int e) constexpr -> int { return PAIRED ? (e >> 1) * (2 * FPW) + (e & 1) : e * FPW; };
    struct DynBase { T *e, *o; };   // element (B + c), B per lane, c a compile-time constant: one base for even c, one for odd c
    auto dyn = [&](int B) -> DynBase {
        if (!PAIRED) return DynBase{lp + B * FPW, lp + B * FPW};
        const int pb = (B >> 1) * (2 * FPW) + (B & 1);
        return DynBase{lp + pb, lp + ((B & 1) ? ((B + 1) >> 1) * (2 * FPW) : pb + 1)};
    };
    auto at = [&](const DynBase &d, int c) -> T & { if (!PAIRED) return d.e[c * FPW]; return (c & 1) ? d.o[PX(c - 1)] : d.e[PX(c)]; };
    const DynBase dq = dyn(q);       // slot (e + q): own column / own entry of a vector
    T *const lqn = lp + q * NS * FPW;   // slot (e + q n), n even: own row q + L r of a row-major n-column matrix starts at slot (L r) n from here
    auto ep = [&](const T *ubase, int rt, int c) -> gptr { return (gptr)anchored(ubase, rt, c); };
    // (model streams: non-temporal where a lane group reads whole 128-byte segments (L <= 4); with eight lanes per filter a group reads HALF
    // a line and the part next door the other half a little later -- the streaming hint lets the line leave the L2 in between and it comes
    // from memory twice (kb_srif_split.h: 1.36x the packed reads with the hint, 1.04x without), so there the default policy)
    auto ldg = [&](const T *ubase, int rt, int c, unsigned off) { if constexpr (L == 8) return *(ep(ubase, rt, c) + off); else return __builtin_nontemporal_load(ep(ubase, rt, c) + off); };
    bool rowok[RP], rowany[RP];
#pragma unroll
    for (int r = 0; r < RP; r++) { rowok[r] = !GEN || q + L * r < rn; rowany[r] = !GEN || L * r < rn; }
    unsigned utri[RP];   // 64 tri(i_r): packed element (0, i_r); elements (l, i_r), l < i_r, follow
#pragma unroll
    for (int r = 0; r < RP; r++) utri[r] = (unsigned)(((q + L * r) * (q + L * r + 1) / 2) * KB_TILE);
    // own row i_r of a packed symmetric matrix: element (i_r, l) right of the diagonal at tri(l) + i_r, left of it at tri(i_r) + l
    auto sym_off = [&](int r, int l, unsigned base_q, unsigned base) -> unsigned {   // lane offset relative to the field's element 0
        if (l >= L * r + L - 1) return base_q + (unsigned)((l * (l + 1) / 2 + L * r) * KB_TILE);
        if (l < L * r) return base + utri[r] + (unsigned)(l * KB_TILE);
        return l >= q + L * r ? base_q + (unsigned)((l * (l + 1) / 2 + L * r) * KB_TILE) : base + utri[r] + (unsigned)(l * KB_TILE);
    };

    // The distributed pivoted LU solve Bm^T Y = Rm^T (used for Z, and for State(prev) with FULL): column c of the matrix is Bm (own row c
    // of B), column c of the right-hand side is Rm; on return Rm[r][j] = Y[j][i_r].  zero_pivot: some pivot was exactly zero.
    auto solve = [&](T (&Bm)[RP][NS], T (&Rm)[RP][NS], bool &zero_pivot) __attribute__((always_inline)) {
        sfor<0, NS>([&](auto KK) __attribute__((always_inline)) {
            constexpr int k = KK, rk = k / L, qk = k % L;
            {   // the owner of column k: pivot (first largest |entry| among rows k..n-1, LAPACK idamax), reciprocal, multipliers
                int piv = k;
                T best = fabs(Bm[rk][k]);
    #pragma unroll
                for (int r2 = k + 1; r2 < NS; r2++) {
                    const bool gt = fabs(Bm[rk][r2]) > best;
                    best = gt ? fabs(Bm[rk][r2]) : best;
                    piv = gt ? r2 : piv;
                }
                T pv = Bm[rk][k];
    #pragma unroll
                for (int r2 = k + 1; r2 < NS; r2++) pv = (piv == r2) ? Bm[rk][r2] : pv;
                const T rpv = T(1) / pv;
                if (q == qk) {
                    lp[PX(BOFF + 0)] = (T)piv;
                    lp[PX(BOFF + 1)] = pv;
    #pragma unroll
                    for (int r2 = k + 1; r2 < NS; r2++) {
                        // the multiplier of row r2 AFTER the exchange: the entry that sits in row r2 then is the old row k's if r2 == piv
                        const T e = (piv == r2) ? Bm[rk][k] : Bm[rk][r2];
                        lp[PX(BOFF + 1 + r2)] = e * rpv;
                    }
                }
            }
            wave_lds_fence();
            {
                const int piv = (int)lp[PX(BOFF + 0)];
                zero_pivot = zero_pivot || (lp[PX(BOFF + 1)] == T(0));
                T mult[NS];
    #pragma unroll
                for (int r2 = k + 1; r2 < NS; r2++) mult[r2] = lp[PX(BOFF + 1 + r2)];
                // rows k and piv change places in every column: entries k and piv of every own row of B and of M -- only when some filter of the
                // wave pivots.  (sfor, not `#pragma unroll` loops: inside the conditional block those are unrolled too late for the arrays
                // to be promoted to registers: 592 B of scratch per lane.)
                if (__any(piv != k))
                sfor<0, RP>([&](auto RR) __attribute__((always_inline)) {
                    constexpr int r = RR;
                    T bk = Bm[r][k], mk = Rm[r][k];
                    T bp = bk, mp = mk;
                    sfor<k + 1, NS>([&](auto R2) __attribute__((always_inline)) {
                        constexpr int r2 = R2;
                        const bool hit = piv == r2;
                        bp = hit ? Bm[r][r2] : bp;
                        mp = hit ? Rm[r][r2] : mp;
                        Bm[r][r2] = hit ? bk : Bm[r][r2];
                        Rm[r][r2] = hit ? mk : Rm[r][r2];
                    });
                    Bm[r][k] = bp;
                    Rm[r][k] = mp;
                });
    #pragma unroll
                for (int r = 0; r < RP; r++) {
                    const bool right = r > rk || (r == rk && q > qk);   // own column q + L r of the matrix lies right of k
    #pragma unroll
                    for (int r2 = k + 1; r2 < NS; r2++) {
                        if (r >= rk) Bm[r][r2] -= (right ? mult[r2] : T(0)) * Bm[r][k];
                        Rm[r][r2] -= mult[r2] * Rm[r][k];
                    }
                }
    #pragma unroll
                for (int r = 0; r < RP; r++)
    #pragma unroll
                    for (int c = 0; c < NS; c++) { pin(Bm[r][c]); pin(Rm[r][c]); }
            }
            wave_lds_fence();
            KB_SB();
        });
        // back substitution, row i = n - 1 .. 0: U[i][c] for c >= i sits with the owners of columns c (Bm[.][i]); they hand row i over
        sfor<0, NS>([&](auto II) __attribute__((always_inline)) {
            constexpr int i = NS - 1 - II;
    #pragma unroll
            for (int r = 0; r < RP; r++)
                if (L * r + L - 1 >= i) {
                    if (q + L * r >= i) at(dq, BOFF + L * r) = Bm[r][i];   // slot c = q + L r holds U[i][c]
                }
            wave_lds_fence();
            {
                const T rd = T(1) / lp[PX(BOFF + i)];
                T urow[NS];
    #pragma unroll
                for (int c = i + 1; c < NS; c++) urow[c] = lp[PX(BOFF + c)];
    #pragma unroll
                for (int r = 0; r < RP; r++) {
                    T s = Rm[r][i];
    #pragma unroll
                    for (int c = i + 1; c < NS; c++) s -= urow[c] * Rm[r][c];
                    Rm[r][i] = s * rd;
                    pin(Rm[r][i]);
                }
            }
            wave_lds_fence();
            KB_SB();
        });
    };

    // ---- FULL: State(prev) = inverse(I) i (information.go:284-288), yhat = H State(prev) [+ Measurement(k)] (:192-194) -----------------------
    [[maybe_unused]] T *const es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems);
    if constexpr (FULLT) {
        T Ar[RP][NS], Er[RP][NS];   // own rows of I (mirrored; read again in phase 0: cache hits), own rows of the identity
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int l = 0; l < NS; l++) {
                const bool okl = rowok[r] && l < rn, diag = (l % L == q && l / L == r);
                const gptr pe = ep(st, rn, 0) + (okl ? sym_off(r, l, uq, us) : us);
                const T v = (rowany[r] && l < rn) ? *pe : T(0);
                Ar[r][l] = okl ? v : (diag ? T(1) : T(0));   // padding: an identity block
                Er[r][l] = diag ? T(1) : T(0);
            }
        // |I|_inf (mat64's condition test, kb_device.h inverse_lu): the own row sums, gathered, the same running maximum in every lane
        auto gathered_max = [&](const T (&rs)[RP]) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < RP; r++) at(dq, BOFF + L * r) = rs[r];
            wave_lds_fence();
            T m = T(0);
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const T v = lp[PX(BOFF + i)];
                if (i < rn) m = (v > m || v != v) ? v : m;
            }
            wave_lds_fence();
            return m;
        };
        T rs[RP];
#pragma unroll
        for (int r = 0; r < RP; r++) {
            T sacc = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) sacc += fabs(Ar[r][l]);
            rs[r] = sacc;
        }
        const T anorm = gathered_max(rs);
        bool bad = false;
        solve(Ar, Er, bad);
        // Er[r][j] = X[j][i_r]: the own COLUMNS of the inverse.  All of it goes to LDS (slot j n + c), free at this point
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int j = 0; j < NS; j++) at(dq, j * NS + L * r) = Er[r][j];
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < RP; r++) {
            T sacc = T(0);
#pragma unroll
            for (int c = 0; c < NS; c++) sacc += fabs(lqn[PX((L * r) * NS + c)]);
            rs[r] = sacc;
        }
        const T inorm = gathered_max(rs);
        bad = bad || !(anorm * inorm <= T(1e16));
        T ivp[NS];   // the information vector: requested here (and again in phase 0), not carried through the solve
#pragma unroll
        for (int l = 0; l < NS; l++) ivp[l] = l < rn ? *(ep(st, 0, l) + us) : T(0);
        // State(prev)[i_r] = sum_j P[i_r][j] i[j], P mirrored from the upper triangle of X (AsSymDense): X[i_r][j] right of the diagonal
        // (own row: slot i_r n + j), X[j][i_r] left of it (slot j n + i_r)
#pragma unroll
        for (int r = 0; r < RP; r++) {
            T sacc = T(0);
#pragma unroll
            for (int j = 0; j < NS; j++) {
                T pij;
                if (j >= L * r + L - 1) pij = lqn[PX((L * r) * NS + j)];
                else if (j < L * r) pij = at(dq, j * NS + L * r);
                else pij = *(j >= q + L * r ? &lqn[PX((L * r) * NS + j)] : &at(dq, j * NS + L * r));
                sacc += pij * ivp[j];
            }
            rs[r] = bad ? T(0) : sacc;   // (zeros: information.go:286-288)
        }
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < RP; r++) at(dq, BOFF + L * r) = rs[r];
        wave_lds_fence();
        // yhat: lane q forms the measurement rows c = q + L r2 (row c of H, and of chol(R) with AWGN) and stores them
        [[maybe_unused]] T z1[NM];
        const bool awgn = a.noise_kind == KB_NOISE_AWGN;
        if (awgn) {
            const uint64_t gfi = (uint64_t)(a.first_filter + tile * KB_TILE) + (unsigned)slot;
            const uint32_t stepno = (uint32_t)a.step0 - (active ? a.lag[tile * KB_TILE + slot] : 0u);   // kf.step of this filter
            draw_normals<T, NM>(a, gfi, stepno, 1u, z1);
        }
#pragma unroll
        for (int r2 = 0; r2 < PC; r2++) {
            const int cbase = L * r2;
            const bool real = (cbase + L - 1 < NM || q + cbase < NM) && q + cbase < rp;
            const unsigned urow = real ? um + (unsigned)((q + cbase) * rn * KB_TILE) : um;
            T sacc = T(0);
#pragma unroll
            for (int l = 0; l < NS; l++) sacc += ((cbase < rp && l < rn) ? ldg(mo, a.L.mo_H, l, urow) : T(0)) * lp[PX(BOFF + l)];
            if (awgn) {
                const unsigned utr = real ? um + (unsigned)(((q + cbase) * (q + cbase + 1) / 2) * KB_TILE) : um;
                T v = T(0);
#pragma unroll
                for (int i = 0; i < NM; i++) {
                    const bool in = real && i <= q + cbase;
                    const T lr = (cbase < rp && i < cbase + L && i < rp) ? ldg(mo, a.L.mo_LR, i, utr) : T(0);
                    v += (in ? lr : T(0)) * z1[i];
                }
                sacc += v;
            }
            const gptr pe = ep(es, a.L.es_yhat, cbase) + uq;   // (formed outside the lane-dependent branch)
            if (real && active) __builtin_nontemporal_store(sacc, pe);
        }
        wave_lds_fence();
        KB_SB();
    }
    // ---- phase 0: F^-1 (own rows -> LDS), i (all), I (own rows) -------------------------------------------------------------------------
    T iv[NS], Ir[RP][NS];
    {
        T Fr[RP][NS];
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int l = 0; l < NS; l++) {
                const T v = (rowany[r] && l < rn) ? ldg(mo, a.L.mo_Finv + (GEN ? L * r * rn : 0), (GEN ? 0 : L * r * NS) + l, rowok[r] ? uf : um) : T(0);
                Fr[r][l] = rowok[r] ? v : T(0);
            }
        auto load_state = [&](auto NT) {
#pragma unroll
            for (int l = 0; l < NS; l++) {
                const gptr pe = ep(st, 0, l) + us;
                iv[l] = l < rn ? (decltype(NT)::value ? __builtin_nontemporal_load(pe) : *pe) : T(0);
            }
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int l = 0; l < NS; l++) {
                    const bool okl = rowok[r] && l < rn;
                    const gptr pe = ep(st, rn, 0) + (okl ? sym_off(r, l, uq, us) : us);
                    const T v = (rowany[r] && l < rn) ? (decltype(NT)::value ? __builtin_nontemporal_load(pe) : *pe) : T(0);
                    Ir[r][l] = okl ? v : T(0);
                }
        };
        KB_WITH_STATE_POLICY(a, load_state);
        KB_SB();
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int l = 0; l < NS; l++) lqn[PX((L * r) * NS + l)] = Fr[r][l];   // slot i_r n + l
    }
    wave_lds_fence();
    KB_SB();

    // ---- phase A: T1 = I F^-1 (own rows), one row of F^-1 per chunk; the own columns of F^-1; F^-T i ---------------------------------------
    T T1[RP][NS], Fc[RP][NS], im[RP];
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int j = 0; j < NS; j++) T1[r][j] = T(0);
    {
        T row[2][NS];
        auto fetch = [&](int l, int b) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NS; j++) row[b][j] = lp[PX(l * NS + j)];
        };
        fetch(0, 0);
#pragma unroll
        for (int l = 0; l < NS; l++) {
            if (l + 1 < NS) fetch(l + 1, (l + 1) & 1);
            KB_SB();
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int j = 0; j < NS; j++) { T1[r][j] += Ir[r][l] * row[l & 1][j]; pin(T1[r][j]); }
            KB_SB();
        }
    }
#pragma unroll
    for (int r = 0; r < RP; r++) {
        T s = T(0);
#pragma unroll
        for (int l = 0; l < NS; l++) {
            Fc[r][l] = at(dq, l * NS + L * r);   // F^-1[l][i_r]
            s += Fc[r][l] * iv[l];
        }
        im[r] = s;   // (F^-T i)[i_r]
        pin(im[r]);
    }
    KB_SB();
    // ---- phase B: T1 takes F^-1's place in LDS; M = F^-T T1 (own rows), one row of T1 per chunk --------------------------------------------
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int j = 0; j < NS; j++) lqn[PX((L * r) * NS + j)] = T1[r][j];
    wave_lds_fence();
    KB_SB();
    T Mr[RP][NS];   // M[i_r][.]: first as M, then (the right-hand side of the solve) turning into -Z[i_r][.]
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int j = 0; j < NS; j++) Mr[r][j] = T(0);
    {
        T row[2][NS];
        auto fetch = [&](int l, int b) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NS; j++) row[b][j] = lp[PX(l * NS + j)];
        };
        fetch(0, 0);
#pragma unroll
        for (int l = 0; l < NS; l++) {
            if (l + 1 < NS) fetch(l + 1, (l + 1) & 1);
            KB_SB();
#pragma unroll
            for (int r = 0; r < RP; r++)
#pragma unroll
                for (int j = 0; j < NS; j++) { Mr[r][j] += Fc[r][l] * row[l & 1][j]; pin(Mr[r][j]); }
            KB_SB();
        }
    }
    // ---- phase C: B = M + Q^-1 (own rows); M goes to LDS (every row is needed again for I-); the pivoted LU solve ----------------------------
    T Br[RP][NS];
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int c = 0; c < NS; c++) {
            const bool okc = rowok[r] && c < rn;
            const T v = (rowany[r] && c < rn) ? ldg(mo, a.L.mo_Qinv, 0, okc ? sym_off(r, c, umq, um) : um) : T(0);
            Br[r][c] = Mr[r][c] + (okc ? v : ((c % L == q && c / L == r) ? T(1) : T(0)));   // padding: an identity block keeps it invertible
        }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < RP; r++)
#pragma unroll
        for (int j = 0; j < NS; j++) lqn[PX((L * r) * NS + j)] = Mr[r][j];
    wave_lds_fence();
    KB_SB();
    bool ignored = false;   // (the reference ignores this inverse's error, :171)
    solve(Br, Mr, ignored);
    // Mr[r][j] = Y[j][i_r] = (M B^-1)[i_r][j] = -Z[i_r][j]
    // ---- phase D: i- = (1 + Z)(F^-T i [+ M G u]) (:176-185), I- = M + Z M^T (:188-190; own rows, columns j >= L r) ---------------------------
    if constexpr (NC > 0) {
        if (rm > 0) {   // (F^-T i) += M (G u): the own rows of M are read back from LDS
            const T *up = (const T *)a.u + tile * a.u_ts;
            T u[NC], gu[NS];
#pragma unroll
            for (int c = 0; c < NC; c++) u[c] = (active && c < rm) ? ldnt_at(&(up + (int64_t)c * a.u_es)[us]) : T(0);
#pragma unroll
            for (int i = 0; i < NS; i++) {
                T s = T(0);
#pragma unroll
                for (int c = 0; c < NC; c++) s += ((i < rn && c < rm) ? ldg(mo, a.L.mo_G + (GEN ? i * rm : 0), (GEN ? 0 : i * NC) + c, um) : T(0)) * u[c];
                gu[i] = s;
            }
#pragma unroll
            for (int r = 0; r < RP; r++) {
                T s = T(0);
#pragma unroll
                for (int j = 0; j < NS; j++) s += lqn[PX((L * r) * NS + j)] * gu[j];
                im[r] = im[r] + s;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RP; r++) at(dq, BOFF + L * r) = im[r];
    wave_lds_fence();
    T imn[RP];
#pragma unroll
    for (int r = 0; r < RP; r++) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < NS; j++) s += (((j % L == q && j / L == r) ? T(1) : T(0)) + T(-1) * Mr[r][j]) * lp[PX(BOFF + j)];
        imn[r] = s;
        pin(imn[r]);
    }
    T Im[RP][NS];   // [r][j], j >= L r
    {
        T row[2][NS];
        auto fetch = [&](int j, int b) __attribute__((always_inline)) {
#pragma unroll
            for (int l = 0; l < NS; l++) row[b][l] = lp[PX(j * NS + l)];
        };
        fetch(0, 0);
#pragma unroll
        for (int j = 0; j < NS; j++) {
            if (j + 1 < NS) fetch(j + 1, (j + 1) & 1);
            KB_SB();
#pragma unroll
            for (int r = 0; r < RP; r++)
                if (L * r <= j) {
                    T s = T(0);
#pragma unroll
                    for (int l = 0; l < NS; l++) s += (T(-1) * Mr[r][l]) * row[j & 1][l];
                    Im[r][j] = lqn[PX((L * r) * NS + j)] + s;   // M[i_r][j] + (Z M^T)[i_r][j]
                    pin(Im[r][j]);
                }
            KB_SB();
        }
    }
    if constexpr (FULLT) {   // I- leaves at once (Estimate.PredCovariance, information.go:295-316 inverts it lazily), as in kb_information_reg.hip
#pragma unroll
        for (int r = 0; r < RP; r++)
#pragma unroll
            for (int j = L * r; j < NS; j++) {
                const gptr pe = ep(es, a.L.es_ppred, j * (j + 1) / 2 + L * r) + uq;
                if (active && rowok[r] && j < rn && (j >= L * r + L - 1 || j >= q + L * r)) __builtin_nontemporal_store(Im[r][j], pe);
            }
    }
    // ---- phase E: H^T R^-1 (own rows), i+ = H^T R^-1 y + i-, I+ = I- + H^T R^-1 H (:197-212) ------------------------------------------------
    T Hp[NM][RP], Ri[TM], y[NM];
#pragma unroll
    for (int c = 0; c < NM; c++)
#pragma unroll
        for (int r = 0; r < RP; r++) {
            const T v = (rowany[r] && c < rp) ? ldg(mo, a.L.mo_H + (GEN ? c * rn : 0), (GEN ? 0 : c * NS) + L * r, rowok[r] ? umq : um) : T(0);
            Hp[c][r] = rowok[r] ? v : T(0);
        }
#pragma unroll
    for (int c = 0; c < NM; c++)
#pragma unroll
        for (int r2 = 0; r2 <= c; r2++) Ri[symi(r2, c)] = c < rp ? ldg(mo, a.L.mo_Rinv, symi(r2, c), um) : T(0);
    {
        const T *yp = (const T *)a.y + tile * a.y_ts;
#pragma unroll
        for (int c = 0; c < NM; c++) y[c] = (active && c < rp) ? ldnt_at(&(yp + (int64_t)c * a.y_es)[us]) : T(0);
    }
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < NM; c++)
#pragma unroll
        for (int r = 0; r < RP; r++) at(dq, c * NS + L * r) = Hp[c][r];   // H[c][i_r] at slot c n + i_r
    wave_lds_fence();
    KB_SB();
    T HTR[RP][NM], ip[RP];
#pragma unroll
    for (int r = 0; r < RP; r++) {
#pragma unroll
        for (int j = 0; j < NM; j++) {
            T s = T(0);
#pragma unroll
            for (int l = 0; l < NM; l++) s += Hp[l][r] * Ri[symi(l, j)];
            HTR[r][j] = s;
        }
        T s = T(0);
#pragma unroll
        for (int j = 0; j < NM; j++) s += HTR[r][j] * y[j];
        ip[r] = s + imn[r];
    }
    T Ip[RP][NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        T hcol[NM];
#pragma unroll
        for (int l = 0; l < NM; l++) hcol[l] = lp[PX(l * NS + j)];
#pragma unroll
        for (int r = 0; r < RP; r++)
            if (L * r <= j) {
                T s2 = T(0);
#pragma unroll
                for (int l = 0; l < NM; l++) s2 += HTR[r][l] * hcol[l];
                Ip[r][j] = Im[r][j] + s2;
            }
    }
    // ---- non-finite screen, stores ---------------------------------------------------------------------------------------------------------
    unsigned err;
    {
        T chk = T(0);
#pragma unroll
        for (int r = 0; r < RP; r++) {
            chk += ip[r] * T(0);
#pragma unroll
            for (int j = L * r; j < NS; j++) chk += Ip[r][j] * T(0);
        }
        err = sum_lanes<L>((chk != chk) ? (unsigned)KB_ST_NONFINITE : 0u);
    }
    if (active && err == 0) {
        auto store_state = [&](auto NT) {
#pragma unroll
            for (int r = 0; r < RP; r++) {
                if (rowok[r]) {
                    const gptr pe = ep(st, 0, L * r) + uq;
                    if constexpr (decltype(NT)::value) __builtin_nontemporal_store(ip[r], pe); else *pe = ip[r];
                }
#pragma unroll
                for (int j = L * r; j < NS; j++)
                    if (rowok[r] && j < rn && (j >= L * r + L - 1 || j >= q + L * r)) {
                        const gptr pe = ep(st, rn, j * (j + 1) / 2 + L * r) + uq;
                        if constexpr (decltype(NT)::value) __builtin_nontemporal_store(Ip[r][j], pe); else *pe = Ip[r][j];
                    }
            }
        };
        KB_WITH_STATE_POLICY(a, store_state);
    }
    const unsigned lane_end = late_lane();
    if (active && err && ((lane_end / FPW) & (L - 1)) == 0)
        atomicOr(a.status + tile * KB_TILE + (int64_t)((gw % L) * FPW + (lane_end & (FPW - 1))), (unsigned)KB_ST_NONFINITE);
}

template <typename T, int NS, int NM, int NC, int L, bool GEN, bool FULLT = false>
__global__ void __launch_bounds__(64, 2) information_split_kernel(const StepArgs a) {
    __shared__ __attribute__((aligned(16))) T lds[(infsplit_lds_elems<NS>() + 1) / 2 * 2 * (64 / L)];   // (whole pairs)
    information_split_part<T, NS, NM, NC, L, GEN, FULLT>(a, split_part_of_block<L>(blockIdx.x, gridDim.x), lds);   // (kb_vanilla_split.h: XCD-aware for L = 8)
}
#undef KB_SB

}  // namespace kb
