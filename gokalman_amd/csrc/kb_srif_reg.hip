// kb_srif_reg.hip -- SRIF Predict() (srif.go:111-141: the time update alone) with one filter per lane, for the benchmark
// shape (n = 12, p = 6) and the reference tests' shape (n = 6, p = 2), and the SRIF dispatch.  Update() runs in the
// two-lanes-per-filter kernel (kb_srif_pair.h), other shapes in the generic kernel (kb_kinds.hip).
// Phi is read in place from the caller's planar arrays after kb_prepare_dev (zero-copy) or from the model block after
// kb_prepare.  The kernel rewrites (b, R) as (bBar, RBar) in place; RBar = R Phi^-1 is dense, which the next Update's DENSE
// variant picks up (Batch::srif_tri).
// Differences from the statement-by-statement generic kernel (rounding level only): State(prev) = R^-1 b and
// RBar = R Phi^-1 are obtained by LU solves instead of inverse-then-multiply (srif.go:111-115, :223-234); only exact
// singularity / non-finite results are flagged (the generic kernel also applies gonum's cond > 1e16 test).
// Failure semantics, the same on every SRIF path and the reference's (srif.go:111-114: `return nil, err` before anything is
// assigned, kf.step not advanced): a filter whose Phi (or R, in State(prev)) is singular at step k gets KB_ST_SINGULAR and
// keeps its estimate for THAT step only; step k+1 runs normally (the status word is a sticky report, not a gate).
#include "kb_internal.h"
#include "kb_static.h"
#include <cstdlib>

namespace kb {


__device__ __forceinline__ int nib(uint64_t perm, int r) { return (int)((perm >> (4 * r)) & 15u); }

// ---- time update (srif.go:111-141): b <- bBar, R <- RBar = R Phi^-1, in place -------------------
// One wave alone on a SIMD issues a VALU instruction every 4 cycles, two waves every 2 (MI355X_MICROARCH.md,
// "vector-instruction ISSUE cost"), and this kernel is VALU-paced (~4.5k instructions per filter tile), so it is
// written to fit 256 registers and a few KB of LDS per wave: Phi is factorised in VGPRs (row exchanges by
// select, skipped wave-uniformly when no lane needs one), the factors stay there for the 12 row solves, and LDS
// is only the scatter buffer that undoes the row permutation (24 floats per lane).
template <typename T, int NS, bool FULL, bool EXT>
__global__ void __launch_bounds__(256, (sizeof(T) * NS * NS > 600 ? 1 : 2)) srif_time_kernel(const StepArgs a) {
    constexpr int RG = 2;
    __shared__ T lds[4 * RG * NS * KB_TILE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    if (tile >= a.ntiles) return;
    const int64_t fi = tile * KB_TILE + lane;
    const bool active = fi < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (NS + NS * NS)) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *ephi = EXT ? (const T *)a.ext_phi + (active ? fi : 0) : nullptr;
    T *ltmp = lds + wv * (RG * NS * KB_TILE) + lane;
    unsigned err = 0;
    T xprev[NS];
    {   // State(prev) = R^-1 b (srif.go:223-234)
        T Rw[NS * NS];
#pragma unroll
        for (int i = 0; i < NS; i++) xprev[i] = ldt(st, i);
#pragma unroll
        for (int e = 0; e < NS * NS; e++) Rw[e] = ldt(st, NS + e);
        // R is upper triangular whenever the last writer was a measurement update (srif.go:334-337 zeroes the
        // sub-columns) or the constructor (diagonal R0): pivoted LU then degenerates to L = I, U = R, no exchanges.
        // Wave-uniform test; the general path stays for R after Predict() / NON_TRI_R.
        bool lower = false;
#pragma unroll
        for (int i = 1; i < NS; i++)
#pragma unroll
            for (int j = 0; j < i; j++) lower = lower || (Rw[i * NS + j] != T(0));
        if (__any(lower)) {
            if (lu_solve_inplace<T, NS, 1>(Rw, xprev)) err |= KB_ST_SINGULAR;
        } else {
#pragma unroll
            for (int i = NS - 1; i >= 0; i--) {
                T sum = xprev[i];
#pragma unroll
                for (int k2 = i + 1; k2 < NS; k2++) sum -= Rw[i * NS + k2] * xprev[k2];
                if (Rw[i * NS + i] == T(0)) err |= KB_ST_SINGULAR;
                xprev[i] = sum * (T(1) / Rw[i * NS + i]);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    T lu[NS * NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {   // xBar = Phi State(prev) (srif.go:118) -> LDS slots, read back in pivoted order below
        T s = T(0);
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const T v = EXT ? __builtin_nontemporal_load(ephi + (int64_t)(i * NS + j) * a.ext_ld) : ldnt(mo, a.L.mo_F + i * NS + j);
            lu[i * NS + j] = v;
            s += v * xprev[j];
        }
        ltmp[i * KB_TILE] = s;
    }
    // P Phi = L U (srif.go:111-114's Inverse = Dgetrf + ...): nibble k of perm = original index of the row now in position k
    uint64_t perm = 0xFEDCBA9876543210ull;
#pragma unroll
    for (int j = 0; j < NS; j++) {
#pragma unroll
        for (int r = j + 1; r < NS; r++) {
            const bool sw = fabs(lu[r * NS + j]) > fabs(lu[j * NS + j]);
            if (__any(sw)) {
#pragma unroll
                for (int c = 0; c < NS; c++) {  // whole rows: the L part moves with its row (LAPACK dlaswp)
                    const T t0 = lu[j * NS + c], t1 = lu[r * NS + c];
                    lu[j * NS + c] = sw ? t1 : t0;
                    lu[r * NS + c] = sw ? t0 : t1;
                }
                const uint64_t x = sw ? (((perm >> (4 * j)) ^ (perm >> (4 * r))) & 15u) : 0u;
                perm ^= (x << (4 * j)) | (x << (4 * r));
            }
        }
        const T piv = lu[j * NS + j];
        if (piv == T(0)) err |= KB_ST_SINGULAR;
        const T rp = T(1) / piv;
        lu[j * NS + j] = rp;   // the solves multiply by the reciprocal
#pragma unroll
        for (int r = j + 1; r < NS; r++) {
            const T l = lu[r * NS + j] * rp;
            lu[r * NS + j] = l;
#pragma unroll
            for (int c = j + 1; c < NS; c++) lu[r * NS + c] -= l * lu[j * NS + c];
        }
    }
    // failed: (b, R) stay as they are (srif.go:111-114 returns before any assignment and before kf.step++)
    if (err) { if (active) fail_step(a, fi, err); return; }
    T *es = FULL ? (T *)a.est + tile * ((int64_t)KB_TILE * a.L.es_elems) + lane : nullptr;
    int poff[NS];  // LDS element offset of original row perm_k
    T xBarP[NS];   // xBar in pivoted order
#pragma unroll
    for (int r = 0; r < NS; r++) {
        poff[r] = nib(perm, r) * KB_TILE;
        xBarP[r] = ltmp[poff[r]];
    }
    // RBar = R Phi^-1, RG rows at a time (independent dependency chains); the next rows of R are prefetched
    // (L2 hits: the state tile was read a moment ago) while the current ones are solved.
    T znext[RG][NS];
#pragma unroll
    for (int g = 0; g < RG; g++)
#pragma unroll
        for (int l = 0; l < NS; l++) znext[g][l] = ldt(st, NS + g * NS + l);
#pragma unroll 1
    for (int i0 = 0; i0 < NS; i0 += RG) {
        T z[RG][NS];
#pragma unroll
        for (int g = 0; g < RG; g++)
#pragma unroll
            for (int l = 0; l < NS; l++) z[g][l] = znext[g][l];
        if (i0 + RG < NS) {
#pragma unroll
            for (int g = 0; g < RG; g++)
#pragma unroll
                for (int l = 0; l < NS; l++) znext[g][l] = ldt(st, NS + (i0 + RG + g) * NS + l);
        }
        // z Phi = r  with  P Phi = L U:  w U = r,  v L = w,  z[perm_k] = v_k   (srif.go:115)
#pragma unroll
        for (int j = 0; j < NS; j++) {
#pragma unroll
            for (int g = 0; g < RG; g++) {
                T sum = z[g][j];
#pragma unroll
                for (int k2 = 0; k2 < j; k2++) sum -= z[g][k2] * lu[k2 * NS + j];
                z[g][j] = sum * lu[j * NS + j];
            }
        }
#pragma unroll
        for (int j = NS - 2; j >= 0; j--) {
#pragma unroll
            for (int g = 0; g < RG; g++) {
                T sum = z[g][j];
#pragma unroll
                for (int k2 = j + 1; k2 < NS; k2++) sum -= z[g][k2] * lu[k2 * NS + j];
                z[g][j] = sum;
            }
        }
#pragma unroll
        for (int g = 0; g < RG; g++) {
            T bb = T(0);
#pragma unroll
            for (int r = 0; r < NS; r++) {
                bb += z[g][r] * xBarP[r];                        // :119 bBar = RBar xBar (same products, pivoted order)
                ltmp[poff[r] + g * NS * KB_TILE] = z[g][r];     // undo the row permutation through LDS
            }
            if (active) stt(st, i0 + g, bb);  // b <- bBar: rows i0.. of R and their b entries are not read again
        }
        if (active) {
#pragma unroll
            for (int g = 0; g < RG; g++)
#pragma unroll
                for (int c = 0; c < NS; c++) {
                    const T v = ltmp[(g * NS + c) * KB_TILE];
                    stt(st, NS + (i0 + g) * NS + c, v);
                    if constexpr (FULL) stt(es, a.L.es_ppred + (i0 + g) * NS + c, v);
                }
        }
    }
}

static bool srif_shape_ok(const StepArgs &a, int NS, int NM) { return a.n == NS && a.p == NM; }

template <typename T, int NS, int NM>
static bool srif_try_predict(const Batch &b, const StepArgs &a) {
    if (a.n != NS || !a.predict) return false;   // (the time update does not see the measurement: any p)
    const dim3 grid = tile_grid(a.ntiles), block(256);
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0, ext = a.ext_phi != nullptr;
#define KB_T(F_, E_) KB_LAUNCH((srif_time_kernel<T, NS, F_, E_>), grid, block, 0, b.stream, a)
    if (full) { if (ext) KB_T(true, true); else KB_T(true, false); }
    else      { if (ext) KB_T(false, true); else KB_T(false, false); }
#undef KB_T
    return true;
}

// Shapes of the split-lane kernel (kb_srif_split.h, round 5): fp64 everything up to 16 / 8 except 12/6 and 6/2; fp32 what the two-lane kernels do not serve; Update and Predict.  KB_SRIF_SPLIT_ALL=1 (environment, diagnostic) sends every fp64 shape there.
static bool srif_split_all() {
    static const bool on = [] { const char *e = getenv("KB_SRIF_SPLIT_ALL"); return e && *e && *e != '0'; }();
    return on;
}
bool srif_split_ok(const Batch &b, const StepArgs &a) {
    if ((a.flags & KB_FLAG_STATEMENT_KERNELS) || a.n < 1 || a.n > 16 || a.p < 1 || a.p > 8) return false;
    if (b.dtype == KB_F32) {
        // fp32: the two-lane kernels keep the even state dimensions 6 .. 16 (14, 16: Update with p <= 6 only); everything else -- odd n,
        // n < 6 (round 4: widened shadow copies, ~2x the bytes: 7/3 116 us, 11/4 258 us), Predict() and p = 7, 8 at 14 / 16 states
        // (round 4: the statement kernel) -- runs the split kernel on four lanes per filter
        if ((a.n & 1) || a.n < 6) return true;
#ifdef KB_DIAG_SRIF_F32_N12
        if (a.n == 12 && a.p == 6 && !a.predict && srif_split_all()) return true;
#endif
        // (p = 7, 8 at the even n up to 12 stays two-lane: 6/8 45 us against 74 on the split kernel, 8/8 57 / 87, 10/8 83 / 159, 12/8 116 / 172)
        return a.n > 12 && (a.predict || a.p > 6);
    }
    // (measured, 256k filters, us per step, split / two-lane kernel: 12/8 232 / 390, 8/8 140 / 172, 12/5 201 / 234, 12/2 181 / 182, 10/8
    // 201 / 204, 10/6 183 / 177, 10/4 166 / 152, 8/2 83 / 71, 12/6 235 / 214, 6/2 64 / 39.  The two-lane fp64 kernel keeps the benchmark shape
    // 12/6 and the reference tests' 6/2; everything else takes the split kernel -- within 15 % where it loses, and eighteen two-lane
    // instantiations (a third of the library's build time) are gone.)
    return !((a.n == 12 && a.p == 6) || (a.n == 6 && a.p == 2)) || srif_split_all();
}
static int launch_srif_split(const Batch &b, const StepArgs &a) {
    typedef void (*launch_t)(const Batch &, const StepArgs &);
    static const launch_t by_n[17] = {nullptr, launch_srif_split_n1, launch_srif_split_n2, launch_srif_split_n3, launch_srif_split_n4, launch_srif_split_n5,
                                      launch_srif_split_n6, launch_srif_split_n7, launch_srif_split_n8, launch_srif_split_n9, launch_srif_split_n10,
                                      launch_srif_split_n11, launch_srif_split_n12, launch_srif_split_n13, launch_srif_split_n14, launch_srif_split_n15,
                                      launch_srif_split_n16};
    static const launch_t by_n32[17] = {nullptr, launch_srif_split_f32_n1, launch_srif_split_f32_n2, launch_srif_split_f32_n3, launch_srif_split_f32_n4,
                                        launch_srif_split_f32_n5, nullptr, launch_srif_split_f32_n7, nullptr, launch_srif_split_f32_n9, nullptr,
                                        launch_srif_split_f32_n11,
#ifdef KB_DIAG_SRIF_F32_N12
                                        launch_srif_split_f32_n12,
#else
                                        nullptr,
#endif
                                        launch_srif_split_f32_n13, launch_srif_split_f32_n14, launch_srif_split_f32_n15,
                                        launch_srif_split_f32_n16};
    (b.dtype == KB_F32 ? by_n32 : by_n)[a.n](b, a);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

// zero-copy Phi / Htilde (kb_prepare_dev) need one of the kernels that read the caller's planar arrays
bool srif_reg_ok(const Batch &b, const StepArgs &a) {
    if (a.flags & KB_FLAG_STATEMENT_KERNELS) return false;
    if (srif_split_ok(b, a)) return true;
    if (!a.predict && a.ext_ld >= (int64_t(1) << 28)) return false;   // the two-lane kernel's 32-bit byte offsets (kb_srif_pair.h)
    if (a.n == 14 || a.n == 16) return b.dtype == KB_F32 && !a.predict && a.p >= 1 && a.p <= 6;   // (fp32, Update only: kb_srif_pair32f.hip ...)
    if (a.n != 6 && a.n != 8 && a.n != 10 && a.n != 12) return false;
    return a.p >= 1 && a.p <= 8;
}

int launch_srif(const Batch &b, const StepArgs &a) {
    if (a.flags & KB_FLAG_STATEMENT_KERNELS) return launch_srif_gen(b, a);
    if (srif_split_ok(b, a)) return launch_srif_split(b, a);   // kb_srif_split.h
    bool done = false;
    if (!a.predict) done = b.dtype == KB_F32 ? (launch_srif_pair_f32(b, a) || launch_srif_pair_f32b(b, a) || launch_srif_pair_f32c(b, a) || launch_srif_pair_f32d(b, a) || launch_srif_pair_f32e(b, a) || launch_srif_pair_f32f(b, a) || launch_srif_pair_f32g(b, a))
                                             : launch_srif_pair_f64(b, a);   // kb_srif_pair.h
    else if (b.dtype == KB_F32) done = srif_try_predict<float, 12, 6>(b, a) || srif_try_predict<float, 6, 2>(b, a) || srif_try_predict<float, 8, 2>(b, a) || srif_try_predict<float, 8, 4>(b, a) ||
                                       srif_try_predict<float, 10, 2>(b, a) || srif_try_predict<float, 10, 4>(b, a) || srif_try_predict<float, 12, 2>(b, a) || srif_try_predict<float, 12, 4>(b, a);
    else done = srif_try_predict<double, 6, 2>(b, a) || srif_try_predict<double, 12, 6>(b, a);
    if (!done) return launch_srif_gen(b, a);
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
