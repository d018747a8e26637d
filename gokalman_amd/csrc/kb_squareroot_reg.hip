// kb_squareroot_reg.hip -- register-resident SquareRoot step (squareroot.go:129-274) for the
// benchmark shapes: dimensions are template parameters, both QR panels live in VGPRs, and the
// structural zeros of the panels (sqrtQ^T / sqrtR^T upper-triangular blocks) are skipped at
// compile time.  Same one-filter-per-lane / AoSoA-64 mapping as kb_vanilla.hip; the model
// (F, H, chol(Q), chol(R)) and the measurements are streamed with non-temporal loads so that
// x and S stay resident in the Infinity Cache across steps.
//
// Per filter-step it reads x[n], S (packed lower), F[n^2], H[p n], chol(Q), chol(R) (packed), y[p]
// and writes x, S: the same 1488 algorithmic bytes as Vanilla with S, sqrtQ, sqrtR in place of
// P, Q, R.
#include "kb_internal.h"
#include "kb_static.h"
#include "kb_vanilla_reg.h"   // draw_normals / chol_times / TilePtr (the AWGN draws of the register kernels)

namespace kb {
#ifndef SQRT_WPB
#define SQRT_WPB 1   // waves per workgroup
#endif
// Waves per SIMD the register allocation is held to: the exact 6/3 fp64 step fits 256 registers (242, no scratch) once the
// panel builds are pinned below; the padded 6/4 shapes (10 x 10 second panel) do not.
template <typename T, int NS, int NM, bool FULL, bool PAD>
constexpr int sqrt_waves() { return (sizeof(T) == 8 && NS > 4 && PAD) ? 1 : 2; }


template <int NS>
struct ActC {  // C = [S^T F^T ; sqrtQ^T]: bottom block row r' is non-zero in column k only when r' <= k
    static constexpr bool active(int k, int r) { return r < NS || (r - NS) <= k; }
};
template <int NM>
struct ActD {  // Delta: sqrtR^T is upper triangular, so rows k+1..NM-1 of its column k are zero
    static constexpr bool active(int k, int r) { return k >= NM || r >= NM; }
};

// PAD: run-time dimensions a.n <= NS, a.p <= NM, a.m <= NC on operands padded with zeros (and an identity block in
// chol(R)): zero rows / columns of a QR panel produce no reflection (sqr_r's `refl` test) and leave the real entries
// untouched, so only loads and stores see the real sizes (cf. kb_vanilla_reg.h).
// NOISE: the batch's Noise is AWGN (noise.go:109-164).  SquareRoot.Update draws twice per step (squareroot.go:239, :268):
// Measurement(k) into yhat -- only a FULL estimate keeps it -- and Process(k) into x+; x- carries no noise (:139-147).  The
// draws come at the very end, when only x+, S+ (and K) are alive, with the stream indices of the generic kernel (which = 1, 2).
// SHARED: instantiated for batches with ONE model for all filters (StepArgs::mo_ts == 0): the model operands are read from lane 0's
// copy in tile 0's block with the default cache policy -- wave-uniform addresses, scalar loads where no store precedes them
// (kb_vanilla_reg.h ldm)
// FUSED (round 5): the caller loop `for k { kf.Update(y_k) }` inside one launch (kb_update_steps_dev): x, S and the model (F, H, chol Q,
// chol R: 81 doubles at 6 / 3) stay in registers over a.nsteps steps, one wave per SIMD; the one-step kernel's source in a loop, with the
// divisions of the two factorisations and of the 3 x 3 inverse replaced by Newton-refined reciprocals (the kernel is bound by instruction
// issue: 13.9 -> 15.4 G filter-steps/s, bench.py extra.squareroot.fused).  Not PROMISED to be the one-step kernel's bits (a Newton-refined
// reciprocal is within an ulp of the quotient, not always equal to it) and held to 1e-12 against T launches and to the oracle like them
// (tests/test_kinds_gpu.py) -- but since round 6 it IS bit-identical on every batch tried (1000 filters x 6 steps, scripts/diag_sqrt_fused.py):
// the last-place differences of round 5 came from the compiler contracting `u0 a + x y` in sqr_r() one way round in the one-step
// instantiation and the other way round in this one; sqr_r now spells its fmas out (kb_static.h).
// Noiseless, state only, no control.
#ifndef KB_SQRT_FUSED_FASTDIV
#define KB_SQRT_FUSED_FASTDIV 1   // (0: diagnostic builds, scripts/diag_sqrt_fused.py -- the time-fused kernel with the one-step kernel's divisions)
#endif
template <typename T, int NS, int NM, int NC, bool FULL, bool PAD = false, bool NOISE = false, bool SHARED = false, bool FUSED = false>
__global__ void __launch_bounds__(64 * SQRT_WPB, (FUSED ? 1 : sqrt_waves<T, NS, NM, FULL, PAD>())) squareroot_reg_kernel(const StepArgs a) {
    static_assert(!FUSED || (!FULL && !PAD && !NOISE && !SHARED && NC == 0), "the time-fused variant: Noiseless, state only, exact shape");
    constexpr int TR = tri(NS), TM = tri(NM), DD = NS + NM;
    const int rn = PAD ? a.n : NS, rp = PAD ? a.p : NM, rm = PAD ? a.m : NC;
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * SQRT_WPB + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;
    const bool active = tile * KB_TILE + lane < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (rn + tri(rn))) + lane;
    // one model for all filters (StepArgs::mo_ts == 0): every lane reads lane 0's copy in tile 0's block -- 8 bytes per load, not a 512-byte row
    const T *mo = SHARED ? (const T *)a.model : (const T *)a.model + tile * a.mo_ts + (a.mo_ts ? lane : 0);
    auto ldmo = [&](const T *q, int e) __attribute__((always_inline)) { return SHARED ? q[(int64_t)e * KB_TILE] : ldnt(q, e); };
    const T *yp = (const T *)a.y + tile * a.y_ts + lane;

    // request order "slowest first" (kb_vanilla_reg.h): F is an HBM stream, x and S are Infinity-Cache hits
    T x[NS], S[TR], F[NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) F[i * NS + j] = (i < rn && j < rn) ? ldmo(mo, a.L.mo_F + i * rn + j) : T(0);
    auto load_state = [&](auto NT) {   // cache policy of the state block: kb_vanilla_reg.h
        constexpr bool nt = decltype(NT)::value;
#pragma unroll
        for (int i = 0; i < NS; i++) x[i] = (i < rn) ? ldp<nt>(st, i) : T(0);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int k2 = 0; k2 <= i; k2++) S[symi(k2, i)] = (i < rn) ? ldp<nt>(st, rn + symi(k2, i)) : T(0);  // S[i][k], k <= i, at symi(k, i)
    };
    KB_WITH_STATE_POLICY(a, load_state);
    [[maybe_unused]] T Hres[FUSED ? NM * NS : 1], LQres[FUSED ? TR : 1], LRres[FUSED ? TM : 1];
    if constexpr (FUSED) {
#pragma unroll
        for (int e = 0; e < NM * NS; e++) Hres[e] = ldmo(mo, a.L.mo_H + e);
#pragma unroll
        for (int e = 0; e < TR; e++) LQres[e] = ldmo(mo, a.L.mo_LQ + e);
#pragma unroll
        for (int e = 0; e < TM; e++) LRres[e] = ldmo(mo, a.L.mo_LR + e);
    }
    __builtin_amdgcn_sched_barrier(0);

    const int nsteps = FUSED ? a.nsteps : 1;
    unsigned bad = 0;
    for (int t = 0; t < nsteps; t++) {
    // :139-147 x- = F x [+ G u]
    T xm[NS];
    smv<T, NS, NS>(F, x, xm);
    if constexpr (NC > 0) {
        const T *up = (const T *)a.u + tile * a.u_ts + lane;
#pragma unroll
        for (int i = 0; i < NS; i++) {
            T s = T(0);
#pragma unroll
            for (int c = 0; c < NC; c++)
                if (i < rn && c < rm) s += ldmo(mo, a.L.mo_G + i * rm + c) * (active ? __builtin_nontemporal_load(up + (int64_t)c * a.u_es) : T(0));
            xm[i] = xm[i] + s;
        }
    }
    // :155-185 C = [S^T F^T ; sqrtQ^T] -> Uc; QUIRK S- := Uc (upper)
    // Build order bounds the live set: the top block row by row (column i of S dies with row i), then a
    // scheduling barrier, and only then chol(Q) for the bottom block -- F is dead by then.  Without the
    // barrier the compiler hoists every load to the top and x, S, F, C and chol(Q) (126 doubles) are
    // alive together.
    T C[2 * NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) {
            T s = T(0);
#pragma unroll
            for (int l = i; l < NS; l++) s += S[symi(i, l)] * F[j * NS + l];  // S[l][i], l >= i
            C[i * NS + j] = s;
        }
    // The reflector's `refl` test splits the factorisation into basic blocks, and LLVM's machine sinking then moves the
    // products above INTO the factorisation (each column of C formed where it is first used), which keeps S and F alive
    // next to the panel: +100 registers.  An empty asm that "modifies" each value pins it before the barrier.
#pragma unroll
    for (int i = 0; i < NS * NS; i++) pin(C[i]);
#pragma unroll
    for (int i = 0; i < NS; i++) pin(xm[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j < NS; j++) C[(NS + i) * NS + j] = (j >= i && j < rn) ? (FUSED ? LQres[symi(i, j)] : ldmo(mo, a.L.mo_LQ + symi(i, j))) : T(0);  // sqrtQ^T[i][j] = L[j][i]
    sqr_r<T, 2 * NS, NS, ActC<NS>, FUSED && KB_SQRT_FUSED_FASTDIV>(C);
    __builtin_amdgcn_sched_barrier(0);  // H, chol(R) loads and the Delta panel stay below the C phase
    // Sm[i][j] = C[i*NS+j], j >= i
    T H[NM * NS];
#pragma unroll
    for (int r = 0; r < NM; r++)
#pragma unroll
        for (int l = 0; l < NS; l++) H[r * NS + l] = (r < rp && l < rn) ? (FUSED ? Hres[r * NS + l] : ldmo(mo, a.L.mo_H + r * rn + l)) : T(0);
    // :190-216 Delta = [[sqrtR^T, 0],[S-^T H^T, S-^T]]
    T D[DD * DD];
#pragma unroll
    for (int r = 0; r < DD; r++)
#pragma unroll
        for (int c = 0; c < DD; c++) {
            T val;
            if (c < NM) {
                if (r < NM) {
                    val = (c >= r) ? (c < rp ? (FUSED ? LRres[symi(r, c)] : ldmo(mo, a.L.mo_LR + symi(r, c))) : (r == c ? T(1) : T(0))) : T(0);
                } else {
                    T s = T(0);  // (S-^T H^T)[r-NM][c] = sum_{l <= r-NM} Sm[l][r-NM] H[c][l]
#pragma unroll
                    for (int l = 0; l <= r - NM; l++) s += C[l * NS + (r - NM)] * H[c * NS + l];
                    val = s;
                }
            } else if (r < NM) {
                val = T(0);
            } else {
                val = (c - NM <= r - NM) ? C[(c - NM) * NS + (r - NM)] : T(0);  // S-^T[r-NM][c-NM] = Sm[c-NM][r-NM]
            }
            D[r * DD + c] = val;
        }
    // H x- for the innovation (:255-262) is formed here, so that H (18 doubles) is dead before the 9 x 9 factorisation
    T Hxm[NM];
    smv<T, NM, NS>(H, xm, Hxm);
    if constexpr (FULL) {
        // :237-239 yhat = H x_prev.  x_prev is read a second time here (an L2 hit) instead of being kept in registers
        // through the first factorisation; yhat and Uc leave at once.  (With NOISE yhat is formed at the end, next to its draw.)
        asm volatile("" ::: "memory");
        [[maybe_unused]] T xp[NS], yhat[NM];
        if constexpr (!NOISE) {
#pragma unroll
            for (int i = 0; i < NS; i++) xp[i] = (i < rn) ? ldt(st, i) : T(0);
            smv<T, NM, NS>(H, xp, yhat);
        }
        T *es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane;
        if (active) {
#pragma unroll
            for (int i = 0; i < NS; i++)
#pragma unroll
                for (int j = i; j < NS; j++)
                    if (j < rn) stnt(es, a.L.es_ppred + symi(i, j), C[i * NS + j]);
            if constexpr (!NOISE) {
#pragma unroll
                for (int r = 0; r < NM; r++)
                    if (r < rp) stnt(es, a.L.es_yhat + r, yhat[r]);
            }
        }
    }
#pragma unroll
    for (int r = NM; r < DD; r++)
#pragma unroll
        for (int c = 0; c < NM; c++) pin(D[r * DD + c]);
#pragma unroll
    for (int r = 0; r < NM; r++) pin(Hxm[r]);
    __builtin_amdgcn_sched_barrier(0);
    sqr_r<T, DD, DD, ActD<NM>, FUSED && KB_SQRT_FUSED_FASTDIV>(D);
    // :225-252 Syy = UD[:p,:p]^T, W = UD[:p,p:]^T, K = W Syy^-1 (general inverse, error ignored)
    T Syy[NM * NM], SyyI[NM * NM], K[NS * NM];
#pragma unroll
    for (int i = 0; i < NM; i++)
#pragma unroll
        for (int j = 0; j < NM; j++) Syy[i * NM + j] = (j <= i) ? D[j * DD + i] : T(0);
    inverse_lu<T, NM, FUSED && KB_SQRT_FUSED_FASTDIV>(Syy, SyyI, rp);
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int c = 0; c < NM; c++) {
            T s = T(0);
#pragma unroll
            for (int k2 = 0; k2 < NM; k2++) s += D[k2 * DD + (NM + i)] * SyyI[k2 * NM + c];  // W[i][k2] = UD[k2][p+i]
            K[i * NM + c] = s;
        }
    // :255-268
    T innov[NM], xn[NS];
#pragma unroll
    for (int r = 0; r < NM; r++) {
        const T yv = (active && r < rp) ? __builtin_nontemporal_load(yp + (int64_t)t * a.y_step + (int64_t)r * a.y_es) : T(0);
        innov[r] = yv - Hxm[r];
    }
    T chk = T(0);
#pragma unroll
    for (int i = 0; i < NS; i++) {
        T s = T(0);
#pragma unroll
        for (int c = 0; c < NM; c++) s += K[i * NM + c] * innov[c];
        xn[i] = xm[i] + s;
    }
    if constexpr (NOISE) {
        // pin: keeps the factorisation and the gain in front of the (wave-uniform) noise branch -- see kb_vanilla_reg.h
#pragma unroll
        for (int i = 0; i < NS; i++) pin(xn[i]);
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) pin(D[(NM + j) * DD + (NM + i)]);
        if constexpr (FULL) {
#pragma unroll
            for (int e = 0; e < NS * NM; e++) pin(K[e]);
#pragma unroll
            for (int r = 0; r < NM; r++) pin(innov[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const TilePtr<const T> mot{(const T *)a.model + tile * a.mo_ts, a.mo_ts ? (unsigned)lane : 0u};
        const uint64_t gfi = (uint64_t)(a.first_filter + tile * KB_TILE) + lane;
        const uint32_t stepno = (uint32_t)a.step0 - (active ? a.lag[tile * KB_TILE + lane] : 0u);   // kf.step of this filter
        T z2[NS], w[NS];
        draw_normals<T, NS>(a, gfi, stepno, 2u, z2);
        chol_times<T, NS>(mot.field(a.L.mo_LQ), rn, z2, w);
#pragma unroll
        for (int i = 0; i < NS; i++) xn[i] += w[i];                      // squareroot.go:268 Process(k)
        if constexpr (FULL) {
            T z1[NM], v[NM], xp[NS], yhat[NM];
            draw_normals<T, NM>(a, gfi, stepno, 1u, z1);
            chol_times<T, NM>(mot.field(a.L.mo_LR), rp, z1, v);
#pragma unroll
            for (int i = 0; i < NS; i++) xp[i] = (i < rn) ? ldt(st, i) : T(0);   // x_prev: the state block is rewritten below
#pragma unroll
            for (int r = 0; r < NM; r++) {                               // squareroot.go:237-239 yhat = H x_prev + Measurement(k)
                T s = T(0);
#pragma unroll
                for (int l = 0; l < NS; l++) s += ((r < rp && l < rn) ? ldmo(mo, a.L.mo_H + r * rn + l) : T(0)) * xp[l];
                yhat[r] = s + v[r];
            }
            T *es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane;
            if (active) {
#pragma unroll
                for (int r = 0; r < NM; r++)
                    if (r < rp) stnt(es, a.L.es_yhat + r, yhat[r]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NS; i++) chk += xn[i] * T(0);
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) chk += D[(NM + j) * DD + (NM + i)] * T(0);
    const bool ok = !(chk != chk);
    if constexpr (FUSED) {
        // a non-finite step leaves (x, S) as they were -- the one-step kernel's predicated store -- and the next step runs normally
#pragma unroll
        for (int i = 0; i < NS; i++) x[i] = ok ? xn[i] : x[i];
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) S[symi(j, i)] = ok ? D[(NM + j) * DD + (NM + i)] : S[symi(j, i)];
        bad |= ok ? 0u : 1u;
        continue;
    }
    if (active && ok) {
        auto store_state = [&](auto NT) {
            constexpr bool nt = decltype(NT)::value;
#pragma unroll
            for (int i = 0; i < NS; i++)
                if (i < rn) stp<nt>(st, i, xn[i]);
#pragma unroll
            for (int i = 0; i < NS; i++)
#pragma unroll
                for (int j = 0; j <= i; j++)
                    if (i < rn) stp<nt>(st, rn + symi(j, i), D[(NM + j) * DD + (NM + i)]);  // S+[i][j] = UD[p+j][p+i]
        };
        KB_WITH_STATE_POLICY(a, store_state);
        if constexpr (FULL) {
            T *es = (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane;
#pragma unroll
            for (int i = 0; i < NS; i++)
#pragma unroll
                for (int c = 0; c < NM; c++)
                    if (i < rn && c < rp) stnt(es, a.L.es_gain + i * a.pmax + c, K[i * NM + c]);
#pragma unroll
            for (int r = 0; r < NM; r++)
                if (r < rp) stnt(es, a.L.es_innov + r, innov[r]);
        }
    }
    if (active && !ok) atomicOr(a.status + tile * KB_TILE + lane, (unsigned)KB_ST_NONFINITE);
    }
    if constexpr (FUSED) {
        if (active) {
#pragma unroll
            for (int i = 0; i < NS; i++) stt(st, i, x[i]);
#pragma unroll
            for (int e = 0; e < TR; e++) stt(st, NS + e, S[e]);
            if (bad) atomicOr(a.status + tile * KB_TILE + lane, (unsigned)KB_ST_NONFINITE);
        }
    }
}

template <typename T, int NS, int NM, int NC = 0, bool NOISE = false, bool SHARED = false>
static bool sqrt_try(const Batch &b, const StepArgs &a) {
    if (a.n != NS || a.p != NM || a.sqrt_p != NM || (a.need_ctrl ? a.m : 0) != NC || a.nsteps != 1) return false;
    if (SHARED && a.mo_ts != 0) return false;
    if (a.noise_kind != (NOISE ? KB_NOISE_AWGN : KB_NOISE_NOISELESS)) return false;
    const dim3 grid((unsigned)((a.ntiles + SQRT_WPB - 1) / SQRT_WPB)), block(64 * SQRT_WPB);
    if (a.flags & KB_FLAG_FULL_ESTIMATE) KB_LAUNCH((squareroot_reg_kernel<T, NS, NM, NC, true, false, NOISE, SHARED>), grid, block, 0, b.stream, a);
    else KB_LAUNCH((squareroot_reg_kernel<T, NS, NM, NC, false, false, NOISE, SHARED>), grid, block, 0, b.stream, a);
    return true;
}

// any (n, p, m) with n <= NS, p <= NM, m <= NC (NC == 0 iff no control input) on the padded instantiation
template <typename T, int NS, int NM, int NC, bool NOISE = false, bool SHARED = false>
static bool sqrt_try_pad(const Batch &b, const StepArgs &a) {
    const int m = a.need_ctrl ? a.m : 0;
    if (a.n > NS || a.p > NM || a.sqrt_p != a.p || m > NC || (NC == 0) != (m == 0) || a.nsteps != 1) return false;
    if (SHARED && a.mo_ts != 0) return false;
    if (a.noise_kind != (NOISE ? KB_NOISE_AWGN : KB_NOISE_NOISELESS)) return false;
    const dim3 grid((unsigned)((a.ntiles + SQRT_WPB - 1) / SQRT_WPB)), block(64 * SQRT_WPB);
    if (a.flags & KB_FLAG_FULL_ESTIMATE) KB_LAUNCH((squareroot_reg_kernel<T, NS, NM, NC, true, true, NOISE, SHARED>), grid, block, 0, b.stream, a);
    else KB_LAUNCH((squareroot_reg_kernel<T, NS, NM, NC, false, true, NOISE, SHARED>), grid, block, 0, b.stream, a);
    return true;
}

// a time-fused register kernel exists for this batch (kb_update_steps_dev keeps x, S and the model in registers over the steps)
bool squareroot_fused_ok(const Batch &b, const StepArgs &a) {
    if (a.flags & (KB_FLAG_STATEMENT_KERNELS | KB_FLAG_FULL_ESTIMATE | KB_FLAG_STRICT_SYMCHECK)) return false;
    return b.dtype == KB_F64 && a.n == 6 && a.p == 3 && a.sqrt_p == 3 && (a.need_ctrl ? a.m : 0) == 0 && a.noise_kind == KB_NOISE_NOISELESS && a.mo_ts != 0;
}

int launch_squareroot(const Batch &b, const StepArgs &a, bool fused) {
    if (a.flags & KB_FLAG_STATEMENT_KERNELS) return launch_squareroot_gen(b, a);
    if (fused && squareroot_fused_ok(b, a)) {
        KB_LAUNCH((squareroot_reg_kernel<double, 6, 3, 0, false, false, false, false, true>), dim3((unsigned)((a.ntiles + SQRT_WPB - 1) / SQRT_WPB)), dim3(64 * SQRT_WPB), 0, b.stream, a);
        KB_HIP(hipGetLastError());
        return KB_OK;
    }
    bool done = false;
    if (b.dtype == KB_F64 && a.mo_ts == 0)   // one model for all filters: the SHARED instantiations (Noiseless; the benchmark shapes and the padded families)
        done = sqrt_try<double, 6, 3, 0, false, true>(b, a) || sqrt_try<double, 4, 2, 0, false, true>(b, a) || sqrt_try_pad<double, 4, 2, 0, false, true>(b, a) ||
               sqrt_try_pad<double, 4, 2, 2, false, true>(b, a) || sqrt_try_pad<double, 6, 4, 0, false, true>(b, a) || sqrt_try_pad<double, 6, 4, 2, false, true>(b, a);
    if (!done && b.dtype == KB_F64)
        done = sqrt_try<double, 6, 3>(b, a) || sqrt_try<double, 4, 2>(b, a) ||
               sqrt_try<double, 4, 1, 1>(b, a) || sqrt_try<double, 4, 2, 1>(b, a);  // examples/jerkcar: 1- and 2-row H, one control
    else if (b.dtype != KB_F64) done = sqrt_try<float, 6, 3>(b, a);
    if (!done && b.dtype == KB_F64)   // shapes without an exact instantiation: padded register kernels up to 6 / 4 / 2
        done = sqrt_try_pad<double, 4, 2, 0>(b, a) || sqrt_try_pad<double, 4, 2, 2>(b, a) || sqrt_try_pad<double, 6, 4, 0>(b, a) ||
               sqrt_try_pad<double, 6, 4, 2>(b, a);
    if (!done && b.dtype == KB_F64)   // AWGN batches: the benchmark shape exactly, everything else up to 6 / 4 / 2 padded
        done = sqrt_try<double, 6, 3, 0, true>(b, a) || sqrt_try_pad<double, 4, 2, 0, true>(b, a) || sqrt_try_pad<double, 4, 2, 2, true>(b, a) ||
               sqrt_try_pad<double, 6, 4, 0, true>(b, a) || sqrt_try_pad<double, 6, 4, 2, true>(b, a);
    if (!done) done = launch_squareroot_split12(b, a) || launch_squareroot_split16(b, a);   // 6 < n <= 16 (p <= 8, m <= 2): one filter over four / eight lanes, kb_squareroot_split.h
    if (!done) return launch_squareroot_gen(b, a);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
