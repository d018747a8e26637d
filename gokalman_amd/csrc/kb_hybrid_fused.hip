// kb_hybrid_fused.hip -- HybridKF, the caller loop `for k { kf.Prepare(Phi_k, Htilde_k); kf.Update(real_k, computed_k) }` (hybrid.go:78-95; the
// statOD ensembles of configs[3] D(ii)) inside ONE launch (kb_update_nl_steps_dev, round 6): one filter per lane as kb_hybrid_reg.h, exact
// 6 / 1, 6 / 2, 6 / 3, CKF and EKF, fp64, zero-copy Phi / Htilde / observations with a step stride, no SNC, state-only outputs.
// x, P stay in registers from step to step and are still STORED every step: memory stays current, so a step that fails for some filter
// (singular innovation covariance, non-finite result: its stores are predicated, its kf.step stands still, as in the one-step kernel)
// simply makes the wave reload at the top of the next step.  R is read once.  Per step the launch moves 656 B per filter at 6 / 2 instead of
// 872.  The arithmetic is kb_hybrid_reg_step.inc, the text the one-step kernel compiles.
#include "kb_hybrid_reg.h"

namespace kb {

template <typename T, int NS, int NM, bool EKF>
__global__ void __launch_bounds__(64, 2) hybrid_fused_kernel(const StepArgs a) {
    [[maybe_unused]] constexpr bool FULL = false, EXT = true, SNCP = false, PAD = false;   // (what kb_hybrid_reg_step.inc asks for)
    constexpr int TR = tri(NS);
    const int rn = NS, rp = NM;
    const int lane = threadIdx.x & 63;
    const int64_t tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    const bool active = tile * KB_TILE + lane < a.N;
    T *st = (T *)a.state + tile * ((int64_t)KB_TILE * (rn + tri(rn))) + lane;
    const T *mo = (const T *)a.model + tile * ((int64_t)KB_TILE * a.L.mo_elems) + lane;
    const T *yr = (const T *)a.y + tile * a.y_ts + lane;
    const T *yc = (const T *)a.y2 + tile * a.y2_ts + lane;
    const int64_t fi = tile * KB_TILE + lane;
    const T *ephi = (const T *)a.ext_phi + (active ? fi : 0);
    const T *eh = (const T *)a.ext_h + (active ? fi : 0);
    T x[NS], P[TR], R[tri(NM)];
#pragma unroll
    for (int c = 0; c < NM; c++)
#pragma unroll
        for (int r = 0; r <= c; r++) R[symi(r, c)] = ldnt(mo, a.L.mo_R + symi(r, c));
    // (measured alternative, round 6: nothing stored until the last step, a failed filter keeping its old x, P by a select -- P then stays alive
    // through the step: 256 registers + 48..152 B of scratch, 96-98 us per 1M-filter step against 101-102; not kept)
    bool reload = true;
    for (int t = 0; t < a.nsteps; t++) {
        {   // (per-step addresses opaque to loop-invariant code motion: kb_srif_pair.h)
            unsigned long long p0 = (unsigned long long)st, p1 = (unsigned long long)ephi, p2 = (unsigned long long)eh, p3 = (unsigned long long)yr, p4 = (unsigned long long)yc;
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4));
            st = (T *)p0; ephi = (const T *)p1; eh = (const T *)p2; yr = (const T *)p3; yc = (const T *)p4;
        }
        T F[NS * NS], H[NM * NS], real[NM], yv[NM];
#pragma unroll
        for (int i = 0; i < NS; i++)
#pragma unroll
            for (int j = 0; j < NS; j++) F[i * NS + j] = __builtin_nontemporal_load(ephi + (int64_t)(i * rn + j) * a.ext_ld);
#pragma unroll
        for (int r = 0; r < NM; r++)
#pragma unroll
            for (int l = 0; l < NS; l++) H[r * NS + l] = __builtin_nontemporal_load(eh + (int64_t)(r * rn + l) * a.ext_ld);
#pragma unroll
        for (int r = 0; r < NM; r++) {
            real[r] = active ? __builtin_nontemporal_load(yr + (int64_t)r * a.y_es) : T(0);
            const T cv = active ? __builtin_nontemporal_load(yc + (int64_t)r * a.y2_es) : T(0);
            yv[r] = real[r] - cv;
        }
        if (reload) {   // the first step, or some filter of this wave failed the last one (its registers hold what the failed step made of them)
            auto load_state = [&](auto NT) {   // cache policy of the state block: kb_vanilla_reg.h (a block beyond the Infinity Cache is streamed)
                constexpr bool nt = decltype(NT)::value;
#pragma unroll
                for (int i = 0; i < NS; i++) x[i] = ldp<nt>(st, i);
#pragma unroll
                for (int j = 0; j < NS; j++)
#pragma unroll
                    for (int i = 0; i <= j; i++) P[symi(i, j)] = ldp<nt>(st, rn + symi(i, j));
            };
            KB_WITH_STATE_POLICY(a, load_state);
        }
        __builtin_amdgcn_sched_barrier(0);
#include "kb_hybrid_reg_step.inc"
        if (active && !err) {
            auto store_state = [&](auto NT) {
                constexpr bool nt = decltype(NT)::value;
#pragma unroll
                for (int i = 0; i < NS; i++) stp<nt>(st, i, xn[i]);
#pragma unroll
                for (int j = 0; j < NS; j++)
#pragma unroll
                    for (int i = 0; i <= j; i++) stp<nt>(st, rn + symi(i, j), Pn[symi(i, j)]);
            };
            KB_WITH_STATE_POLICY(a, store_state);
        }
        if (active && err) fail_step(a, fi, err);   // hybrid.go:150-152 returns before kf.step++
#pragma unroll
        for (int i = 0; i < NS; i++) x[i] = xn[i];
#pragma unroll
        for (int e = 0; e < TR; e++) P[e] = Pn[e];
        reload = __any(err != 0);
        ephi += a.ext_phi_step; eh += a.ext_h_step; yr += a.y_step; yc += a.y2_step;
    }
}

template <int NM>
static bool fused_try(const Batch &b, const StepArgs &a) {
    if (a.p != NM) return false;
    const dim3 grid((unsigned)a.ntiles), block(64);
    if (a.ekf) KB_LAUNCH((hybrid_fused_kernel<double, 6, NM, true>), grid, block, 0, b.stream, a);
    else KB_LAUNCH((hybrid_fused_kernel<double, 6, NM, false>), grid, block, 0, b.stream, a);
    return true;
}

bool launch_hybrid_fused(const Batch &b, const StepArgs &a) {
    if (b.dtype != KB_F64 || a.n != 6 || !a.ext_phi || a.snc || a.predict) return false;
    if (a.flags & (KB_FLAG_FULL_ESTIMATE | KB_FLAG_STRICT_SYMCHECK | KB_FLAG_STATEMENT_KERNELS)) return false;
    return fused_try<2>(b, a) || fused_try<1>(b, a) || fused_try<3>(b, a);
}

}  // namespace kb
