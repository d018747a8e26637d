// kb_api_nl.hip -- C ABI, second half: the NLDKF interface (kalman.go:51-60) for SRIF and
// Hybrid batches, the Noise selection (noise.go) and the Monte-Carlo fan-out (montecarlo.go).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "kb_internal.h"

using namespace kb;

namespace kb {
}  // namespace kb

static int ready_nl(kb_batch *b) {
    if (!b) { set_error("null batch"); return KB_ERR_INVALID; }
    if (!b->initialized) { set_error("kb_init has not been called"); return KB_ERR_INVALID; }
    if (b->kind != KB_SRIF && b->kind != KB_HYBRID && b->kind != KB_BATCH_LS) { set_error("not an NLDKF batch (SRIF / Hybrid / BatchKF)"); return KB_ERR_INVALID; }
    return use_device(*b);
}

static int launch_nl(kb_batch *b, const StepArgs &a) {
    begin_kernel_record();
    const int rc = b->kind == KB_BATCH_LS ? launch_batch_ls(*b, a) : (b->kind == KB_SRIF ? launch_srif(*b, a) : launch_hybrid(*b, a));
    end_kernel_record(*b);
    return rc;
}

extern "C" {

// Prepare(Phi, Htilde): srif.go:82-86, hybrid.go:78-82
int kb_prepare(kb_batch *b, const double *phi, const double *htilde, int64_t count, int broadcast) {
    int rc = ready_nl(b);
    if (rc) return rc;
    if (!phi || !htilde) { set_error("null argument"); return KB_ERR_INVALID; }
    if ((rc = kb_set(b, KB_F, phi, count, broadcast, 0))) return rc;
    if ((rc = kb_set(b, KB_H, htilde, count, broadcast, b->pmax))) return rc;
    b->ext_phi = b->ext_h = nullptr;
    b->locked = 0;
    return KB_OK;
}

int kb_prepare_dev(kb_batch *b, const void *phi, const void *htilde, int64_t ld) {
    int rc = ready_nl(b);
    if (rc) return rc;
    if (!phi || !htilde) { set_error("null argument"); return KB_ERR_INVALID; }
    if (ld < b->N) { set_error("ld < N"); return KB_ERR_INVALID; }
    // zero-copy: the step kernel reads the caller's planar arrays directly (they must stay valid
    // until the update has run); shapes without a register kernel get them packed at launch.
    b->ext_phi = phi; b->ext_h = htilde; b->ext_ld = ld;
    b->have[KB_F] = b->have[KB_H] = true;
    b->locked = 0;
    return KB_OK;
}

// PreparePNT(Gamma): hybrid.go:86-89 (a no-op for SRIF, srif.go:79)
int kb_prepare_pnt(kb_batch *b, const double *gamma, int64_t count, int broadcast) {
    int rc = ready_nl(b);
    if (rc) return rc;
    if (b->kind != KB_HYBRID) return KB_OK;
    if (!gamma) { set_error("null argument"); return KB_ERR_INVALID; }
    if (b->m <= 0) { set_error("batch was created with q = 0: no SNC"); return KB_ERR_INVALID; }
    if (!b->have[KB_Q]) { set_error("SNC needs the process noise Q (kb_set KB_Q)"); return KB_ERR_INVALID; }
    if ((rc = kb_set(b, KB_G, gamma, count, broadcast, 0))) return rc;
    b->snc = 1;
    return KB_OK;
}

static int nl_common(kb_batch *b, StepArgs &a, bool predict) {
    int rc;
    if (b->locked) { set_error("kf is locked (call Prepare() first)"); return KB_ERR_LOCKED; }
    a.predict = predict ? 1 : 0;
    if (b->ext_phi) {
        a.ext_phi = b->ext_phi; a.ext_h = b->ext_h; a.ext_ld = b->ext_ld;
        const bool reg = b->kind == KB_SRIF ? srif_reg_ok(*b, a) : (b->kind == KB_HYBRID ? hybrid_reg_ok(*b, a) : false);
        if (!reg) {  // generic kernel: materialise the model block first
            if ((rc = kb_set_dev(b, KB_F, b->ext_phi, b->ext_ld, 0))) return rc;
            if ((rc = kb_set_dev(b, KB_H, b->ext_h, b->ext_ld, b->pmax))) return rc;
            b->ext_phi = b->ext_h = nullptr;
            a.ext_phi = a.ext_h = nullptr;
        }
    }
    if (b->kind == KB_SRIF && !predict) {
        // Batch::srif_leftover: once the stream has drained, the pinned word tells whether a filter failed in the dense kernel
        if (b->srif_leftover && hipStreamQuery(b->stream) == hipSuccess) after_sync(*b);
        a.srif_leftover = b->srif_leftover;
        // every launch of the dense kernel counts its failures afresh: the word stays up only while some filter still fails in it
        if (!b->srif_tri || b->srif_leftover) KB_HIP(hipMemsetAsync(b->d_srif_fail, 0, sizeof(uint32_t), b->stream));
    }
    if ((rc = launch_nl(b, a))) return rc;
    if (b->kind == KB_SRIF && !predict && !b->srif_tri) b->srif_leftover = 1;   // until a drained stream shows that nobody failed in it
    b->step++;
    b->calls++;
    b->srif_tri = predict ? 0 : 1;  // Predict() leaves the full RBar in R (srif.go:134-141), an Update a triangular R_k
    b->snc = 0;     // hybrid.go:201
    b->locked = 1;  // srif.go:158, hybrid.go:202
    return KB_OK;
}

// The reference returns from a failed step before `kf.sncEnabled = false; kf.locked = true` (hybrid.go:150-152 against :201-202,
// srif.go:112-114 against :157-158): the filter stays prepared and the caller may retry.  A one-filter batch (the drop-in
// use) sees its failed-step count from the host after the synchronisation and keeps that behaviour.
static void nl_after_sync(kb_batch *b, uint32_t lag_before, int snc_before) {
    if (b->N == 1 && b->h_lag && b->h_lag[0] != lag_before) { b->locked = 0; b->snc = snc_before; }
}

// Update(realObservation, computedObservation): srif.go:90-92, hybrid.go:93-95
static int update_nl_host(kb_batch *b, const double *real_obs, int real_rows, const double *computed_obs, int computed_rows,
                          int64_t first, int64_t count, kb_estimate_view *view) {
    int rc = ready_nl(b);
    if (rc) return rc;
    if (b->locked) { set_error("kf is locked (call Prepare() first)"); return KB_ERR_LOCKED; }
    if (!real_obs || !computed_obs) { set_error("null observation"); return KB_ERR_INVALID; }
    if (real_rows != computed_rows) {  // checkMatDims(..., rowsAndcols), srif.go:106, hybrid.go:109
        set_error("dimensions must agree: real observation(%dx1) computed observation(%dx1)", real_rows, computed_rows);
        return KB_ERR_DIMS;
    }
    if (real_rows != b->p) {
        set_error("dimensions must agree: observation(%dx1) Htilde(%dx...)", real_rows, b->p);
        return KB_ERR_DIMS;
    }
    const void *rtile = nullptr, *ctile = nullptr;
    if ((rc = stage_host_vec(*b, real_obs, real_rows, &b->d_y, 0, &rtile))) return rc;
    if ((rc = stage_host_vec(*b, computed_obs, computed_rows, &b->d_y2, 2, &ctile))) return rc;
    StepArgs a;
    fill_step_args(*b, a);
    a.y = rtile; a.y_es = KB_TILE; a.y_ts = (int64_t)KB_TILE * real_rows;
    a.y2 = ctile; a.y2_es = KB_TILE; a.y2_ts = (int64_t)KB_TILE * real_rows;
    const uint32_t lag_before = b->h_lag ? b->h_lag[0] : 0u;
    const int snc_before = b->snc;
    if ((rc = nl_common(b, a, false))) return rc;
    if (view) rc = kb_get_estimate(b, first, count, view);   // synchronises once, for the step and the snapshot
    if (!view || rc) KB_HIP(hipStreamSynchronize(b->stream));
    nl_after_sync(b, lag_before, snc_before);
    return rc;
}

int kb_update_nl(kb_batch *b, const double *real_obs, int real_rows, const double *computed_obs, int computed_rows) {
    return update_nl_host(b, real_obs, real_rows, computed_obs, computed_rows, 0, 0, nullptr);
}
int kb_update_nl_estimate(kb_batch *b, const double *real_obs, int real_rows, const double *computed_obs, int computed_rows,
                          int64_t first, int64_t count, kb_estimate_view *view) {
    if (!view) { set_error("null argument"); return KB_ERR_INVALID; }
    return update_nl_host(b, real_obs, real_rows, computed_obs, computed_rows, first, count, view);
}

int kb_update_nl_dev(kb_batch *b, const void *real_obs, const void *computed_obs, int64_t ld) {
    int rc = ready_nl(b);
    if (rc) return rc;
    if (!real_obs || !computed_obs) { set_error("null observation"); return KB_ERR_INVALID; }
    if (ld < b->N) { set_error("ld < N"); return KB_ERR_INVALID; }
    StepArgs a;
    fill_step_args(*b, a);
    a.y = real_obs; a.y_es = ld; a.y_ts = KB_TILE;
    a.y2 = computed_obs; a.y2_es = ld; a.y2_ts = KB_TILE;
    return nl_common(b, a, false);
}

// The caller loop `for k { kf.Prepare(Phi_k, Htilde_k); kf.Update(real_k, computed_k) }` from one call (round 6).  SRIF 12 / 6 fp32 in the
// steady state (config E's shape) and HybridKF 6 / 1..3 fp64 (configs[3] D(ii)): ONE launch, the state resident in registers between the
// steps (kb_srif_pair.h FUSED, kb_hybrid_fused.hip); every other batch: nsteps Prepare + Update launches back to back on the handle's stream.
int kb_update_nl_steps_dev(kb_batch *b, const void *phi, const void *htilde, int64_t ld, int64_t phi_step, int64_t htilde_step,
                           const void *real_obs, const void *computed_obs, int64_t ld_obs, int64_t obs_step, int nsteps) {
    int rc = ready_nl(b);
    if (rc) return rc;
    if (!phi || !htilde || !real_obs || !computed_obs) { set_error("null argument"); return KB_ERR_INVALID; }
    if (ld < b->N || ld_obs < b->N) { set_error("ld < N"); return KB_ERR_INVALID; }
    if (nsteps < 1) { set_error("nsteps must be >= 1"); return KB_ERR_INVALID; }
    if (b->kind == KB_BATCH_LS) { set_error("BatchKF has no multi-step update"); return KB_ERR_UNSUPPORTED; }
    const size_t w = b->esize();
    if (b->kind == KB_SRIF && b->dtype == KB_F32 && ((b->n == 12 && b->p == 6) || (b->n == 6 && b->p == 2)) && nsteps > 1 && !(b->flags & (KB_FLAG_FULL_ESTIMATE | KB_FLAG_STATEMENT_KERNELS))) {
        if (b->srif_leftover && hipStreamQuery(b->stream) == hipSuccess) after_sync(*b);
        if (b->srif_tri && !b->srif_leftover) {
            StepArgs a;
            fill_step_args(*b, a);
            a.ext_phi = phi; a.ext_h = htilde; a.ext_ld = ld; a.ext_phi_step = phi_step; a.ext_h_step = htilde_step;
            a.y = real_obs; a.y_es = ld_obs; a.y_ts = KB_TILE; a.y_step = obs_step;
            a.y2 = computed_obs; a.y2_es = ld_obs; a.y2_ts = KB_TILE; a.y2_step = obs_step;
            a.nsteps = nsteps;
            begin_kernel_record();
            const bool done = launch_srif_pair_f32_fused(*b, a);
            end_kernel_record(*b);
            if (done) {
                KB_HIP(hipGetLastError());
                b->step += nsteps; b->calls += nsteps;
                b->ext_phi = (const char *)phi + (size_t)(nsteps - 1) * (size_t)phi_step * w;   // what the last Prepare() of the loop left
                b->ext_h = (const char *)htilde + (size_t)(nsteps - 1) * (size_t)htilde_step * w; b->ext_ld = ld;
                b->have[KB_F] = b->have[KB_H] = true;
                b->srif_tri = 1; b->snc = 0; b->locked = 1;
                return KB_OK;
            }
        }
    }
    if (b->kind == KB_HYBRID && b->dtype == KB_F64 && b->n == 6 && b->p >= 1 && b->p <= 3 && nsteps > 1 && !b->snc) {
        StepArgs a;
        fill_step_args(*b, a);
        a.ext_phi = phi; a.ext_h = htilde; a.ext_ld = ld; a.ext_phi_step = phi_step; a.ext_h_step = htilde_step;
        a.y = real_obs; a.y_es = ld_obs; a.y_ts = KB_TILE; a.y_step = obs_step;
        a.y2 = computed_obs; a.y2_es = ld_obs; a.y2_ts = KB_TILE; a.y2_step = obs_step;
        a.nsteps = nsteps; a.predict = 0;
        begin_kernel_record();
        const bool done = launch_hybrid_fused(*b, a);
        end_kernel_record(*b);
        if (done) {
            KB_HIP(hipGetLastError());
            b->step += nsteps; b->calls += nsteps;
            b->ext_phi = (const char *)phi + (size_t)(nsteps - 1) * (size_t)phi_step * w;
            b->ext_h = (const char *)htilde + (size_t)(nsteps - 1) * (size_t)htilde_step * w; b->ext_ld = ld;
            b->have[KB_F] = b->have[KB_H] = true;
            b->snc = 0; b->locked = 1;
            return KB_OK;
        }
    }
    for (int t = 0; t < nsteps; t++) {
        if ((rc = kb_prepare_dev(b, (const char *)phi + (size_t)t * (size_t)phi_step * w, (const char *)htilde + (size_t)t * (size_t)htilde_step * w, ld))) return rc;
        if ((rc = kb_update_nl_dev(b, (const char *)real_obs + (size_t)t * (size_t)obs_step * w, (const char *)computed_obs + (size_t)t * (size_t)obs_step * w, ld_obs))) return rc;
    }
    return KB_OK;
}

// Predict(): srif.go:96-98, hybrid.go:99-101
static int predict_nl_host(kb_batch *b, int64_t first, int64_t count, kb_estimate_view *view) {
    int rc = ready_nl(b);
    if (rc) return rc;
    if (b->kind == KB_BATCH_LS) { set_error("BatchKF has no Predict()"); return KB_ERR_UNSUPPORTED; }
    StepArgs a;
    fill_step_args(*b, a);
    const uint32_t lag_before = b->h_lag ? b->h_lag[0] : 0u;
    const int snc_before = b->snc;
    if ((rc = nl_common(b, a, true))) return rc;
    if (view) rc = kb_get_estimate(b, first, count, view);
    if (!view || rc) KB_HIP(hipStreamSynchronize(b->stream));
    nl_after_sync(b, lag_before, snc_before);
    return rc;
}

int kb_predict_nl(kb_batch *b) { return predict_nl_host(b, 0, 0, nullptr); }
int kb_predict_nl_estimate(kb_batch *b, int64_t first, int64_t count, kb_estimate_view *view) {
    if (!view) { set_error("null argument"); return KB_ERR_INVALID; }
    return predict_nl_host(b, first, count, view);
}

// SmoothAll(estimates) (hybrid.go:209-238, srif.go:165-192; estimates without SNC)
int kb_smooth_all_dev(kb_batch *b, const void *phis, int64_t ld, int steps, void *x_out, void *P_out) {
    int rc = ready_nl(b);
    if (rc) return rc;
    if (!phis || !x_out || !P_out) { set_error("null argument"); return KB_ERR_INVALID; }
    if (ld < b->N) { set_error("ld < N"); return KB_ERR_INVALID; }
    if (steps != kb_step(b)) {  // hybrid.go:210-212
        set_error("incorrect number of estimates provided: %d instead of expected %lld", steps, (long long)kb_step(b));
        return KB_ERR_INVALID;
    }
    const int n = b->n;
    if (b->kind == KB_SRIF) {
        if ((rc = ensure_xp(*b))) return rc;
        void *tmp = b->d_xp;
        rc = launch_materialise(*b, b->d_state, false, tmp);
        if (!rc) rc = launch_smooth(*b, tmp, n + tri(n), 0, n, phis, ld, steps, x_out, P_out);
        hipError_t e = hipStreamSynchronize(b->stream);
        if (!rc && e != hipSuccess) rc = hip_fail(e, "kb_smooth_all_dev");
        return rc;
    }
    return launch_smooth(*b, b->d_state, b->L.st_elems, b->L.st_vec, b->L.st_mat, phis, ld, steps, x_out, P_out);
}

// ---- noise (noise.go) -------------------------------------------------------------------
int kb_set_noise_kind(kb_batch *b, int noise_kind, uint64_t seed) {
    if (!b) { set_error("null batch"); return KB_ERR_INVALID; }
    if (noise_kind == KB_NOISE_BATCH) { set_error("use kb_set_batch_noise to select BatchNoise"); return KB_ERR_INVALID; }
    if (noise_kind != KB_NOISE_NOISELESS && noise_kind != KB_NOISE_AWGN) { set_error("unknown noise kind %d", noise_kind); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    b->noise_kind = noise_kind;
    b->seed = seed;
    if (b->initialized && noise_kind == KB_NOISE_AWGN) {  // NewAWGN(Q, R) panics on non-PD input (noise.go:148-156)
        int not_pd = 0, np2 = 0;
        if ((rc = launch_refresh(*b, KB_Q, &not_pd))) return rc;
        if ((rc = launch_refresh(*b, KB_R, &np2))) return rc;
        if (not_pd + np2) { set_error("process / measurement noise invalid: not positive definite"); return KB_ERR_NOT_PD; }
    }
    return KB_OK;
}

// BatchNoise (noise.go:67-106)
int kb_set_batch_noise(kb_batch *b, const double *process, int nproc, const double *measurement, int nmeas) {
    if (!b || !process || !measurement || nproc < 1 || nmeas < 1) { set_error("bad argument"); return KB_ERR_INVALID; }
    if (b->kind != KB_VANILLA && b->kind != KB_VANILLA_PREDICT) { set_error("BatchNoise needs a Vanilla filter: its noise matrices are zero (noise.go:88-98), which SquareRoot cannot factorise and Information cannot invert"); return KB_ERR_UNSUPPORTED; }
    int rc = use_device(*b);
    if (rc) return rc;
    const int n = b->n, p = b->p;
    auto upload = [&](const double *src, size_t cnt, void **dst) -> int {
        if (*dst) KB_HIP(dev_free(*dst));
        *dst = nullptr;
        KB_HIP(dev_alloc(dst, cnt * b->esize()));
        if (b->dtype == KB_F64) {
            KB_HIP(hipMemcpy(*dst, src, cnt * sizeof(double), hipMemcpyHostToDevice));
        } else {
            std::vector<float> tmp(cnt);
            for (size_t i = 0; i < cnt; i++) tmp[i] = (float)src[i];
            KB_HIP(hipMemcpy(*dst, tmp.data(), cnt * sizeof(float), hipMemcpyHostToDevice));
        }
        return KB_OK;
    };
    if ((rc = upload(process, (size_t)nproc * n, &b->d_bn_proc))) return rc;
    if ((rc = upload(measurement, (size_t)nmeas * p, &b->d_bn_meas))) return rc;
    b->bn_nproc = nproc; b->bn_nmeas = nmeas; b->bn_p = p;
    b->noise_kind = KB_NOISE_BATCH;
    // BatchNoise.ProcessMatrix / MeasurementMatrix return ZERO matrices (noise.go:89-98): the filter propagates with
    // Q = 0 and R = 0, whatever was given to kb_set before
    const std::vector<double> zeros((size_t)(n > p ? n * n : p * p), 0.0);
    if ((rc = kb_set(b, KB_Q, zeros.data(), 1, 1, 0))) return rc;
    if ((rc = kb_set(b, KB_R, zeros.data(), 1, 1, p))) return rc;
    return KB_OK;
}

// Standard normals behind the AWGN draw (filter, epoch, step, which): out[k], k < n (which 0,2) or p (which 1).
// The noise vector is chol_L(Q or R) * out.
int kb_noise_sample(kb_batch *b, int64_t filter, int64_t epoch, int64_t step, int which, double *out) {
    if (!b || !out) { set_error("null argument"); return KB_ERR_INVALID; }
    if (which < 0 || which > 2) { set_error("which must be 0, 1 or 2"); return KB_ERR_INVALID; }
    const int len = (which == 1) ? b->p : b->n;
    for (int k = 0; k < len; k++)
        out[k] = normal_at(b->seed, (uint64_t)filter, (uint32_t)step, (uint32_t)(epoch * 4 + which), k);
    return KB_OK;
}

double kb_noise_normal(uint64_t seed, int64_t filter, int64_t epoch, int64_t step, int which, int k) {
    return normal_at(seed, (uint64_t)filter, (uint32_t)step, (uint32_t)(epoch * 4 + which), k);
}

// ---- Monte-Carlo (montecarlo.go:92-119) ----------------------------------------------------
int kb_mc_run(kb_batch *b, int steps, const double *controls, int ncontrols, int64_t first_run, double *sums) {
    return kb_mc_run_ex(b, steps, controls, ncontrols, first_run, sums, 0u);
}

int kb_mc_run_ex(kb_batch *b, int steps, const double *controls, int ncontrols, int64_t first_run, double *sums, unsigned mc_flags) {
    if (!b || !sums) { set_error("null argument"); return KB_ERR_INVALID; }
    double *d_folded = nullptr, *d_shift = nullptr;
    int rc = mc_run_device(*b, steps, controls, ncontrols, first_run, mc_flags, &d_folded, &d_shift);
    if (rc) return rc;
    const int n = b->n;
    std::vector<double> host((size_t)steps * 3 * n);   // shift [steps][n] | folded [steps][2][n]: adjacent in Batch::d_mc
    KB_HIP(hipMemcpyAsync(host.data(), d_shift, host.size() * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    KB_HIP(hipStreamSynchronize(b->stream));
    const double *shift = host.data(), *folded = host.data() + (size_t)steps * n;
    for (int t = 0; t < steps; t++)
        for (int i = 0; i < n; i++) {   // sums[steps][3][n]: sum(d), sum(d^2), shift c
            sums[((size_t)t * 3 + 0) * n + i] = folded[((size_t)t * 2 + 0) * n + i];
            sums[((size_t)t * 3 + 1) * n + i] = folded[((size_t)t * 2 + 1) * n + i];
            sums[((size_t)t * 3 + 2) * n + i] = shift[(size_t)t * n + i];
        }
    return KB_OK;
}

}  // extern "C"

int kb::mc_run_device(Batch &bb, int steps, const double *controls, int ncontrols, int64_t first_run, unsigned mc_flags, double **folded, double **shift) {
    kb_batch *b = static_cast<kb_batch *>(&bb);
    if (mc_flags & ~(unsigned)KB_MC_KEEP_RUNS) { set_error("unknown Monte-Carlo flags 0x%x", mc_flags); return KB_ERR_INVALID; }
    if (!b->initialized) { set_error("kb_init has not been called"); return KB_ERR_INVALID; }
    if (b->kind != KB_VANILLA_PREDICT) {  // montecarlo.go:93-95 (a panic there)
        set_error("the Kalman filter needed for the Monte Carlo runs must be a pure predictor");
        return KB_ERR_INVALID;
    }
    if (steps < 1) { set_error("steps must be >= 1"); return KB_ERR_INVALID; }
    if (ncontrols != 1 && ncontrols != steps) {  // montecarlo.go:105-107 (a panic there)
        set_error("must provide as much control vectors as steps, or just one control vector");
        return KB_ERR_INVALID;
    }
    if (b->noise_kind != KB_NOISE_AWGN) { set_error("Monte-Carlo runs need AWGN noise (kb_set_noise_kind)"); return KB_ERR_INVALID; }
    // montecarlo.go:92-119 runs ONE filter `samples` times: the per-step sums are taken about the noise-free trajectory of that filter
    // (the shift the statistics add back), which is only THE trajectory when every run starts from the same x0 with the same model
    if (b->per_filter_model || b->per_filter_init) {
        set_error("Monte-Carlo runs are N copies of one filter: upload x0, P0 and the model with broadcast = 1, or build the batch with kb_replicate");
        return KB_ERR_INVALID;
    }
    int rc = use_device(*b);
    if (rc) return rc;
    const int n = b->n, m = b->m;
    // controls -> device, batch dtype
    if (b->need_ctrl) {
        if (!controls) { set_error("controls required (needCtrl)"); return KB_ERR_INVALID; }
        const size_t cnt = (size_t)ncontrols * m;
        const size_t bytes = cnt * b->esize();
        if (b->ctrl_bytes < bytes) {
            if (b->d_ctrl) KB_HIP(dev_free(b->d_ctrl));
            b->d_ctrl = nullptr; b->ctrl_bytes = 0;
            KB_HIP(dev_alloc(&b->d_ctrl, bytes));
            b->ctrl_bytes = bytes;
        }
        if (b->dtype == KB_F64) {
            KB_HIP(hipMemcpyAsync(b->d_ctrl, controls, bytes, hipMemcpyHostToDevice, b->stream));
            KB_HIP(hipStreamSynchronize(b->stream));
        } else {
            std::vector<float> tmp(cnt);
            for (size_t i = 0; i < cnt; i++) tmp[i] = (float)controls[i];
            KB_HIP(hipMemcpyAsync(b->d_ctrl, tmp.data(), bytes, hipMemcpyHostToDevice, b->stream));
            KB_HIP(hipStreamSynchronize(b->stream));
        }
    }
    const int repl = mc_repl();
    const size_t nrep = (size_t)repl * steps * 2 * n;   // [repl][steps][2][n] | shift [steps][n] | folded [steps][2][n]
    const size_t ndbl = nrep + (size_t)steps * n + (size_t)steps * 2 * n;
    if (b->mc_bytes < ndbl * sizeof(double)) {
        if (b->d_mc) KB_HIP(dev_free(b->d_mc));
        b->d_mc = nullptr; b->mc_bytes = 0;
        KB_HIP(dev_alloc((void **)&b->d_mc, ndbl * sizeof(double)));
        b->mc_bytes = ndbl * sizeof(double);
    }
    KB_HIP(hipMemsetAsync(b->d_mc, 0, ndbl * sizeof(double), b->stream));
    b->mc_epoch = -1;   // a previous trajectory buffer no longer describes "the last Monte-Carlo run"
    void *traj = nullptr;
    const int64_t traj_ld = b->ntiles * KB_TILE;
    if (mc_flags & KB_MC_KEEP_RUNS) {
        // MonteCarloRuns.Runs (montecarlo.go:11-15, :108-117): samples x steps estimates, what AsCSV and NewChiSquare read
        const size_t bytes = (size_t)steps * (size_t)(n + b->p) * (size_t)traj_ld * b->esize();
        if (bytes > (size_t)KB_MC_KEEP_MAX_BYTES) {
            set_error("keeping %lld runs x %d steps needs %.1f GiB on the device, above the %d GiB cap: run without KB_MC_KEEP_RUNS "
                      "(Mean / StdDev / NewChiSquare do not need the trajectories)", (long long)b->N, steps, bytes / 1073741824.0,
                      (int)(KB_MC_KEEP_MAX_BYTES >> 30));
            return KB_ERR_INVALID;
        }
        if (b->traj_bytes < bytes) {
            if (b->d_traj) KB_HIP(dev_free(b->d_traj));
            b->d_traj = nullptr; b->traj_bytes = 0;
            KB_HIP(dev_alloc(&b->d_traj, bytes));
            b->traj_bytes = bytes;
        }
        traj = b->d_traj;
    }
    StepArgs a;
    fill_step_args(*b, a);
    a.nsteps = steps;
    a.first_filter = first_run;
    a.step0 = 0;
    if ((rc = launch_mc(*b, a, b->d_ctrl, ncontrols, b->d_mc, traj, traj_ld))) return rc;
    if (traj) { b->mc_steps = steps; b->mc_p = b->p; b->mc_ld = traj_ld; b->mc_first_run = first_run; b->mc_epoch = b->epoch; }
    *shift = b->d_mc + nrep;
    *folded = b->d_mc + nrep + (size_t)steps * n;
    if ((rc = launch_fold(b->stream, b->d_mc, repl, (int64_t)steps * 2 * n, *folded))) return rc;
    b->epoch++;  // kf.Reset() after the sample (montecarlo.go:116): state untouched, noise re-seeded
    return KB_OK;
}

extern "C" {

// MonteCarloRuns.Runs[first + k].Estimates[t].State() / .Measurement() of the last kb_mc_run_ex(..., KB_MC_KEEP_RUNS)
int kb_mc_get_runs(kb_batch *b, int64_t first, int64_t count, double *states, double *measurements) {
    if (!b || (!states && !measurements)) { set_error("null argument"); return KB_ERR_INVALID; }
    if (!b->d_traj || b->mc_epoch < 0) { set_error("no Monte-Carlo runs kept on this batch: call kb_mc_run_ex with KB_MC_KEEP_RUNS first"); return KB_ERR_INVALID; }
    if (first < 0 || count < 0 || first + count > b->N) { set_error("runs [%lld,+%lld) outside the batch", (long long)first, (long long)count); return KB_ERR_INVALID; }
    if (count == 0) return KB_OK;
    int rc = use_device(*b);
    if (rc) return rc;
    const int n = b->n, p = b->mc_p, steps = b->mc_steps;
    const size_t per_run = (size_t)steps * (size_t)(n + p) * sizeof(double);
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(count, (int64_t)((size_t)256 << 20) / (int64_t)per_run));
    if ((rc = ensure_stage(*b, (size_t)chunk * per_run))) return rc;
    for (int64_t k0 = 0; k0 < count; k0 += chunk) {
        const int64_t c = std::min(chunk, count - k0);
        double *d_states = (double *)b->d_stage, *d_meas = d_states + (size_t)c * steps * n;
        if ((rc = launch_traj_unpack(*b, first + k0, c, states ? d_states : nullptr, measurements ? d_meas : nullptr))) return rc;
        if (states) KB_HIP(hipMemcpyAsync(states + (size_t)k0 * steps * n, d_states, (size_t)c * steps * n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        if (measurements) KB_HIP(hipMemcpyAsync(measurements + (size_t)k0 * steps * p, d_meas, (size_t)c * steps * p * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        KB_HIP(hipStreamSynchronize(b->stream));
    }
    return KB_OK;
}

// N copies of one filter of an initialised batch (model, initial estimate, noise) as a new batch
int kb_replicate(kb_batch *src, int64_t filter, int64_t nfilters, unsigned flags, kb_batch **out) {
    if (!src || !out) { set_error("null argument"); return KB_ERR_INVALID; }
    *out = nullptr;
    if (!src->initialized) { set_error("kb_init has not been called on the source batch"); return KB_ERR_INVALID; }
    if (filter < 0 || filter >= src->N) { set_error("filter %lld outside the source batch", (long long)filter); return KB_ERR_INVALID; }
    const unsigned inherit = KB_FLAG_INFO_FROM_STATE | KB_FLAG_SRIF_NON_TRI_R;
    kb_batch *d = nullptr;
    int rc = kb_create(&d, src->kind, src->n, src->pmax, src->m, nfilters, src->dtype, src->device,
                       (flags & ~inherit) | (src->flags & inherit));
    if (rc) return rc;
    auto fail = [&](int code) { kb_destroy(d); return code; };
    hipError_t e = hipStreamSynchronize(src->stream);   // the source's pending setters have landed
    if (e != hipSuccess) return fail(hip_fail(e, "kb_replicate"));
    if ((rc = launch_replicate(*d, src->d_state0, src->L.st_elems, filter, d->d_state0))) return fail(rc);
    if ((rc = launch_replicate(*d, src->d_state0, src->L.st_elems, filter, d->d_state))) return fail(rc);
    if ((rc = launch_replicate(*d, src->d_model, src->L.mo_elems, filter, d->d_model))) return fail(rc);
    d->p = src->p; d->r_p = src->r_p; d->rinv_p = src->rinv_p; d->sqrt_p = src->sqrt_p; d->need_ctrl = src->need_ctrl;
    for (int i = 0; i < 8; i++) d->have[i] = src->have[i];
    d->noise_kind = src->noise_kind; d->seed = src->seed; d->epoch = src->epoch; d->ekf = src->ekf;
    if (src->noise_kind == KB_NOISE_BATCH) {   // BatchNoise: the recorded vectors are shared by every filter of a batch
        const size_t pb = (size_t)src->bn_nproc * src->n * src->esize(), mb = (size_t)src->bn_nmeas * src->bn_p * src->esize();
        if ((e = dev_alloc(&d->d_bn_proc, pb)) != hipSuccess || (e = dev_alloc(&d->d_bn_meas, mb)) != hipSuccess ||
            (e = hipMemcpyAsync(d->d_bn_proc, src->d_bn_proc, pb, hipMemcpyDeviceToDevice, d->stream)) != hipSuccess ||
            (e = hipMemcpyAsync(d->d_bn_meas, src->d_bn_meas, mb, hipMemcpyDeviceToDevice, d->stream)) != hipSuccess)
            return fail(hip_fail(e, "kb_replicate (BatchNoise)"));
        d->bn_nproc = src->bn_nproc; d->bn_nmeas = src->bn_nmeas; d->bn_p = src->bn_p;
    }
    if ((e = hipStreamSynchronize(d->stream)) != hipSuccess) return fail(hip_fail(e, "kb_replicate"));
    d->initialized = true;
    d->step = 0;
    *out = d;
    return KB_OK;
}

// MonteCarloRuns.Mean / StdDev (montecarlo.go:18-59): stat.Mean, stat.StdDev (unbiased)
int kb_mc_stats(const double *sums, int steps, int n, int64_t runs, double *mean, double *stddev) {
    if (!sums || !mean || !stddev || steps < 1 || n < 1 || runs < 1) { set_error("bad argument"); return KB_ERR_INVALID; }
    for (int t = 0; t < steps; t++)
        for (int i = 0; i < n; i++) {
            const double s1 = sums[((size_t)t * 3 + 0) * n + i], s2 = sums[((size_t)t * 3 + 1) * n + i];
            const double c = sums[((size_t)t * 3 + 2) * n + i];
            mean[(size_t)t * n + i] = c + s1 / (double)runs;
            const double var = runs > 1 ? (s2 - s1 * s1 / (double)runs) / (double)(runs - 1) : NAN;
            stddev[(size_t)t * n + i] = std::sqrt(var > 0.0 || var != var ? var : 0.0);
        }
    return KB_OK;
}

}  // extern "C"
