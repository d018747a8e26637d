/*
 * gokalman_amd.h -- C ABI of the MI355X-native batched Kalman engine.
 *
 * This is the drop-in boundary for gokalman's predict/update hot path.  The
 * reference has no FFI of its own: its seam is the Go interfaces LDKF / NLDKF /
 * Estimate (kalman.go:35-72), Noise (noise.go:13-20) and the constructors.  A
 * cgo shim (go/gokalman_amd.go) implements those interfaces by
 * binding exactly the entry points declared here; each entry point cites the
 * reference method it replaces.
 *
 * One `kb_batch` = N independent filters of one kind and one shape resident in
 * the HBM of one MI355X.  A reference filter object is a batch with N == 1; the
 * reference's caller loop over many filter objects is a batch with N == many.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types.
 *  - HOST arrays are per-filter row-major ("AoS"): a field with E elements per
 *    filter is `[count][E]` (count == N, or 1 when `broadcast` != 0).  Symmetric
 *    matrices (P, Q, R) are passed and returned as FULL n x n row-major; only the
 *    upper triangle is read (mat64.SymDense semantics, helper.go:65-84).
 *  - DEVICE arrays handed to the `*_dev` entry points are planar ("SoA"):
 *    element e of filter i at `ptr[e * ld + i]`, `ld >= N` given by the caller.
 *  - every function returns KB_OK (0) or a negative kb_status; kb_last_error()
 *    gives the thread-local message (for dimension errors, the reference's own
 *    "dimensions must agree: ..." string, helper.go:99-130).
 *  - numerical failures never abort a batch: they set bits in a per-filter
 *    status word (kb_get_status) and leave that filter's estimate untouched, as
 *    the reference does when Update returns (nil, err).
 *  - a handle is not thread-safe (neither is a reference filter); distinct
 *    handles may be driven from distinct threads.  All work is enqueued on the
 *    handle's HIP stream; host-facing calls synchronise that stream before
 *    returning, `*_dev` calls do not.
 *  - there is NO CPU fallback: every entry point that touches filter data
 *    returns KB_ERR_NO_DEVICE when no gfx950 device is visible.
 */
#ifndef GOKALMAN_AMD_H
#define GOKALMAN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KB_MAX_DIM 16 /* max state / measurement / control dimension */

typedef struct kb_batch kb_batch;

/* Filter kinds = the reference's constructors. */
typedef enum {
    KB_VANILLA = 1,         /* NewVanilla               vanilla.go:21-40      */
    KB_VANILLA_PREDICT = 2, /* NewPurePredictorVanilla  vanilla.go:43-62      */
    KB_SQUAREROOT = 3,      /* NewSquareRoot            squareroot.go:21-50   */
    KB_INFORMATION = 4,     /* NewInformation / NewInformationFromState information.go:20-81 */
    KB_SRIF = 5,            /* NewSRIF                  srif.go:14-49         */
    KB_HYBRID = 6,          /* NewHybridKF              hybrid.go:23-34       */
    KB_BATCH_LS = 7         /* NewBatchKF (batch least squares) batch.go:34-38: kb_set(KB_R), kb_init, then per
                             * measurement kb_prepare(Phi, H) + kb_update_nl(real, computed) = SetNextMeasurement (batch.go:41-61);
                             * kb_get(KB_STATE / KB_COVAR) = Solve() (batch.go:64-79); KB_RAW_VEC / KB_RAW_MAT = N / Lambda */
} kb_kind;

typedef enum { KB_F64 = 0, KB_F32 = 1 } kb_dtype;

/* kb_create flags */
#define KB_FLAG_FULL_ESTIMATE 0x1u /* materialise every Estimate member each step (P-, K, innovation, yhat); off = state-only outputs (x+, P+) */
#define KB_FLAG_STRICT_SYMCHECK 0x2u /* compute both triangles and run AsSymDense's |M_ij-M_ji| test (helper.go:75) instead of the non-finite test */
#define KB_FLAG_INFO_FROM_STATE 0x4u /* KB_INFORMATION: X/P given to kb_set are (x0,P0) as in NewInformationFromState, not (i0,I0) */
#define KB_FLAG_STATEMENT_KERNELS 0x10u /* run the run-time-dimension, statement-order kernels (csrc/kb_kinds.hip) even where a register kernel exists: the library's own slow GPU reference path, for validation (tests compare the two) */
#define KB_FLAG_SRIF_NON_TRI_R 0x8u  /* NewSRIF(nonTriR = true) (srif.go:13); both settings run the same arithmetic, see srif.go:121-132 */

typedef enum {
    KB_OK = 0,
    KB_ERR_INVALID = -1,     /* bad argument / call order                      */
    KB_ERR_DIMS = -2,        /* "dimensions must agree: ..." (helper.go:99-130) */
    KB_ERR_NO_DEVICE = -3,   /* no MI355X visible; there is no CPU fallback     */
    KB_ERR_HIP = -4,         /* a HIP runtime call failed                       */
    KB_ERR_UNSUPPORTED = -5, /* shape/kind/dtype combination has no kernel      */
    KB_ERR_LOCKED = -6,      /* "kf is locked (call Prepare() first)" srif.go:102, hybrid.go:105 */
    KB_ERR_NOT_PD = -7       /* Cholesky of a non positive definite matrix at construction / SetNoise */
} kb_status;

/* Per-filter status bits (sticky until kb_reset / kb_clear_status). */
#define KB_ST_SINGULAR 0x1u   /* inverse failed or cond > 1e16: vanilla.go:164-167, hybrid.go:150-152, srif.go:112-114 */
#define KB_ST_ASYMMETRIC 0x2u /* AsSymDense failure: vanilla.go:207-215, information.go:214-222 (a panic there)        */
#define KB_ST_NONFINITE 0x4u  /* NaN/Inf reached the estimate                                                          */
#define KB_ST_NYQUIST 0x10u   /* kb_van_loan: "Nyquist sampling criterion not fulfilled" (c2d.go:26-28); F, Q are still valid */
#define KB_ST_INFO_NOT_INVERTIBLE 0x8u /* getter-side: information/SRIF matrix not (yet) invertible, covariance reported as zeros (information.go:284-288) */

/* Fields for kb_set / kb_get. */
typedef enum {
    /* inputs (constructor arguments and Set* of kalman.go:35-47) */
    KB_X = 0,        /* x0 [n]       (KB_INFORMATION without INFO_FROM_STATE: i0)  */
    KB_P = 1,        /* P0 [n][n]    (KB_INFORMATION without INFO_FROM_STATE: I0)  */
    KB_F = 2,        /* F  [n][n]    SetStateTransition                            */
    KB_G = 3,        /* G  [n][m]    SetInputControl                               */
    KB_H = 4,        /* H  [p][n]    SetMeasurementMatrix                          */
    KB_Q = 5,        /* Q  [n][n]    Noise.ProcessMatrix  (KB_HYBRID: [q][q])       */
    KB_R = 6,        /* R  [p][p]    Noise.MeasurementMatrix                        */
    /* outputs = the Estimate interface (kalman.go:64-72) */
    KB_STATE = 16,       /* Estimate.State()          [n]                           */
    KB_COVAR = 17,       /* Estimate.Covariance()     [n][n] full symmetric         */
    KB_PRED_COVAR = 18,  /* Estimate.PredCovariance() [n][n] (needs FULL_ESTIMATE, except SQRT/INFO/SRIF which keep it) */
    KB_GAIN = 19,        /* Gain()                    [n][p] (needs FULL_ESTIMATE)  */
    KB_INNOVATION = 20,  /* Estimate.Innovation()     [p]  (INFO/SRIF: the information vector [n], information.go:272, srif.go:237) */
    KB_MEASUREMENT = 21, /* Estimate.Measurement()    [p]  (needs FULL_ESTIMATE)    */
    KB_RAW_VEC = 22,     /* internal vector: x | i | b                              */
    KB_RAW_MAT = 23,     /* internal matrix: P | S | I | R  [n][n]                  */
    KB_RAW_PRED_MAT = 24 /* internal predicted matrix: P- | S- | I- | Rbar          */
} kb_field;

/* ---- lifetime ------------------------------------------------------------- */
/* Allocates device-resident storage for N filters (n states, up to p measurements,
 * m controls; m may be 0) on HIP device `device`.  For KB_HYBRID `m` is the
 * dimension q of the SNC process noise (Gamma is n x q).  Replaces the
 * allocation half of the constructors (vanilla.go:21, squareroot.go:21,
 * information.go:20, srif.go:14, hybrid.go:23). */
int kb_create(kb_batch **out, int kind, int n, int p, int m, int64_t nfilters,
              int dtype, int device, unsigned flags);
void kb_destroy(kb_batch *b);
const char *kb_last_error(void);
const char *kb_version(void);
int kb_device_count(void);
/* Debugging / reporting aid, no counterpart in the reference (kalman.go:35-72 has one Update per filter type; here (kind, n, p, noise, flags)
 * choose among ~150 kernel instantiations): the instantiation(s) the LAST step of this handle was served by, e.g.
 * "vanilla_split_kernel<double, 12, 6, 0, 4, false, false, false>"; two kernels of one step are joined by " + "; "" before the first
 * step.  The string belongs to the handle and is valid until the next kb_last_kernel call on it.  scripts/dispatch_table.py walks the
 * envelope with it (profiles/dispatch_table.md). */
const char *kb_last_kernel(kb_batch *b);

/* ---- model / initial conditions ------------------------------------------- */
/* Host -> device upload of one input field.  `count` = 1 with broadcast != 0
 * (one model for every filter: the reference's usual use) or N.  `p_rows` is
 * the measurement dimension the array is shaped for (KB_H, KB_R; ignored
 * otherwise): it may differ from the creation `p` as long as it is <= it --
 * jerkcar swaps a 1-row and a 2-row H between steps (examples/jerkcar/main.go:141-159).
 * Before kb_init this stages constructor arguments; after kb_init it has the
 * semantics of the matching setter, including the reference's side effects:
 *   KB_F  SetStateTransition: KB_INFORMATION refreshes F^-1 (information.go:117-123)
 *   KB_Q/KB_R  SetNoise: KB_SQUAREROOT recomputes chol(Q), chol(R) (squareroot.go:100-114);
 *              KB_INFORMATION does NOT refresh Q^-1 / R^-1 (information.go:136-138, bug-compatible)
 *   KB_G  SetInputControl: needCtrl is not recomputed (vanilla.go:101-103)
 * Data is converted to the batch dtype on the way in. */
int kb_set(kb_batch *b, int field, const double *host, int64_t count, int broadcast, int p_rows);
/* Same from HBM: planar input in the batch dtype, element e of filter i at src[e*ld + i],
 * full (unpacked) matrices.  Asynchronous on the handle's stream. */
int kb_set_dev(kb_batch *b, int field, const void *src, int64_t ld, int p_rows);

/* Runs the constructor arithmetic on the device (Cholesky of P0/Q/R, F^-1, Q^-1,
 * R^-1, I0 = P0^-1, b0 = R0 x0 ...), checks needCtrl = !IsNil(G) (vanilla.go:39),
 * snapshots the initial estimate for kb_reset and unlocks kb_update. */
int kb_init(kb_batch *b);

/* A new batch of `nfilters` filters, every one a copy of filter `filter` of the initialised batch `src`: same kind, shape,
 * dtype and device, its model (incl. the constructor products), its INITIAL estimate and its noise selection; step 0.
 * This is how a host shim turns the reference's one-filter arguments into device batches without a host round trip:
 * NewMonteCarloRuns(samples, ..., kf) runs `samples` copies of kf (montecarlo.go:108-117 re-uses kf after Reset()),
 * NewChiSquare(kf, runs, ...) `runs` copies of kf (chisquare.go:38-41).  flags: KB_FLAG_FULL_ESTIMATE /
 * KB_FLAG_STRICT_SYMCHECK of the new batch (the kind-defining flags are inherited). */
int kb_replicate(kb_batch *src, int64_t filter, int64_t nfilters, unsigned flags, kb_batch **out);

/* LDKF.Reset (vanilla.go:121-125): restore the initial estimate, step = 0,
 * clear status words, advance the noise stream (AWGN re-seeds on Reset). */
int kb_reset(kb_batch *b);

/* ---- the hot path ----------------------------------------------------------- */
/* LDKF.Update(measurement, control) for every filter of the batch
 * (vanilla.go:128, squareroot.go:129, information.go:153).
 * meas: host [N][meas_rows]; ctrl: host [N][ctrl_rows] or NULL.  meas_rows / ctrl_rows are the
 * lengths of the caller's vectors: a mismatch with H / G returns KB_ERR_DIMS with the
 * reference's message (vanilla.go:129-135); the control is only checked and used when
 * needCtrl (vanilla.go:129). */
int kb_update(kb_batch *b, const double *meas, int meas_rows, const double *ctrl, int ctrl_rows);
/* Same, measurements already in HBM (planar, batch dtype): element e of filter
 * i at meas[e*ld_meas + i].  Asynchronous on the handle's stream (kb_stream: created non-blocking, it does NOT wait for the
 * null stream): the caller's device arrays of every *_dev entry point are read there, so whatever produced them on another
 * stream has to be complete, or ordered before it with an event, when the call is made -- and they stay untouched until the
 * step has run (kb_synchronize, or an event recorded on kb_stream). */
int kb_update_dev(kb_batch *b, const void *meas, int64_t ld_meas,
                  const void *ctrl, int64_t ld_ctrl);
/* The caller loop `for k { kf.Update(y_k, u_k) }` fused into one launch:
 * meas is planar [T][p][ld] on the device (step t at meas + t*p*ld elements).
 * x, P (S) and the model stay in registers across the T steps where a time-fused register kernel exists -- Vanilla 6/3 and 4/2
 * (Noiseless), Vanilla 6/3 with AWGN or BatchNoise drawn inside the launch, SquareRoot 6/3 (Noiseless); fp64, state-only outputs,
 * per-filter models, no control input -- every other kind / shape / noise runs its single-step register kernel T times, back to
 * back on the stream, from this one call.
 * The Noiseless Vanilla and the SquareRoot time-fused kernels are NOT bit-identical to T calls of kb_update_dev (the AWGN / BatchNoise
 * one is: it keeps the one-step kernel's operations).  The Noiseless Vanilla kernel evaluates the Joseph form as
 * P+ = AP - (AP H^T - K R) K^T with AP = P- - K (P- H^T)^T and divides by Newton-refined reciprocals, where the per-step kernel
 * keeps the reference's order of operations (vanilla.go:197-205) and IEEE division; the SquareRoot kernel replaces the divisions of
 * the two factorisations by Newton-refined reciprocals (a reciprocal within an ulp of the quotient: not promised to be T launches' bits,
 * held to 1e-12 of them; in every batch tried since round 6 the bits ARE the same).  All are held to the oracle at 1e-9
 * (tests/test_kinds_gpu.py, bench.py `fused.parity`); on degenerate problems (zero noise matrices) use the per-step call. */
int kb_update_steps_dev(kb_batch *b, const void *meas, int64_t ld_meas,
                        const void *ctrl, int64_t ld_ctrl, int nsteps);

/* NLDKF (kalman.go:51-60).  Prepare(Phi, Htilde) (srif.go:82, hybrid.go:78):
 * host [count][n][n] and [count][p][n]; unlocks the next update. */
int kb_prepare(kb_batch *b, const double *phi, const double *htilde, int64_t count, int broadcast);
int kb_prepare_dev(kb_batch *b, const void *phi, const void *htilde, int64_t ld);
/* PreparePNT(Gamma) (hybrid.go:86-89): enables SNC for the next update only. */
int kb_prepare_pnt(kb_batch *b, const double *gamma, int64_t count, int broadcast);
/* EnableEKF / DisableEKF / EKFEnabled (hybrid.go:48-60). */
int kb_set_ekf(kb_batch *b, int enabled);
int kb_ekf_enabled(const kb_batch *b);
/* NLDKF.Update(realObservation, computedObservation) (srif.go:90, hybrid.go:93). */
int kb_update_nl(kb_batch *b, const double *real_obs, int real_rows, const double *computed_obs, int computed_rows);
int kb_update_nl_dev(kb_batch *b, const void *real_obs, const void *computed_obs, int64_t ld);
/* The caller loop `for k { kf.Prepare(Phi_k, Htilde_k); kf.Update(real_k, computed_k) }` (srif.go:82-92, hybrid.go:78-95) from ONE call:
 * step t reads the planar device arrays (layouts of kb_prepare_dev / kb_update_nl_dev) at phi + t * phi_step, htilde + t * htilde_step,
 * real_obs / computed_obs + t * obs_step (strides in ELEMENTS of the batch's dtype).  SRIF 12 / 6 and 6 / 2 fp32 in the steady state (no Predict()
 * pending) and HybridKF 6 / 1..3 fp64 (CKF and EKF, no SNC pending), state-only outputs, run time-fused kernels -- ONE launch, the state
 * resident in registers from step to step, the same operations in the same order as nsteps single calls: the same bits, kf.step and
 * the per-filter failure semantics included; every other batch runs nsteps Prepare + Update launches back to back on the handle's
 * stream.  Asynchronous like the *_dev calls. */
int kb_update_nl_steps_dev(kb_batch *b, const void *phi, const void *htilde, int64_t ld, int64_t phi_step, int64_t htilde_step,
                           const void *real_obs, const void *computed_obs, int64_t ld_obs, int64_t obs_step, int nsteps);
/* NLDKF.Predict() (srif.go:96, hybrid.go:99). */
int kb_predict_nl(kb_batch *b);

/* SmoothAll(estimates) of HybridKF / SRIF (hybrid.go:209-238, srif.go:165-192; estimates recorded
 * without SNC): backward sweep from the batch's current (last) estimate,
 *   S = inverse(Phi_{k+1});  x_k = S x_{k+1};  P_k = AsSymDense(S P_{k+1} S^T),  k = steps-2 .. 0.
 * phis: the caller's history of the STMs handed to Prepare, planar on the device,
 * phis[(k*n*n + e)*ld + i] = element e of Phi at step k of filter i (the reference keeps Phi inside
 * every estimate; here the caller keeps the arrays it already passed to kb_prepare_dev).
 * `steps` must equal kb_step() ("incorrect number of estimates provided", hybrid.go:210-212).
 * Outputs, planar in the batch dtype: x_out[(k*n + i)*ld + f], P_out[(k*n*n + e)*ld + f] (full n x n).
 * A singular Phi or an asymmetric result stops that filter's sweep and sets its status word. */
int kb_smooth_all_dev(kb_batch *b, const void *phis, int64_t ld, int steps, void *x_out, void *P_out);

/* ---- results ---------------------------------------------------------------- */
/* Device -> host download of one Estimate member for filters [first, first+count),
 * always as float64, host layout.  KB_COVAR of SQRT / INFORMATION / SRIF batches is
 * materialised by a small kernel (S S^T, I^-1, R^-1 R^-T) exactly like the
 * reference's lazy getters (squareroot.go:317, information.go:277, srif.go:253). */
int kb_get(kb_batch *b, int field, double *host, int64_t first, int64_t count);
/* One SNAPSHOT of the Estimate (kalman.go:64-72) of filters [first, first+count): the reference's Update returns a freshly
 * allocated, immutable estimate and keeps it alive (vanilla.go:216-218; consumed later through a channel in
 * examples/jerkcar/main.go:71-90, stored per step in montecarlo.go:108-117).  Every non-NULL member of the view receives
 * its copy (float64, host layout as in kb_get) with one device synchronisation for all of them; NULL members are skipped.
 * pred_covariance / gain / measurement (and innovation of the Vanilla / SquareRoot / Hybrid kinds) need a batch created
 * with KB_FLAG_FULL_ESTIMATE.  status receives the per-filter status bits; with clear_status != 0 the words are read AND
 * cleared in one atomic step, which gives a host shim the reference's per-call error: an Update that fails for a filter
 * leaves that filter's state and covariance as they were and does not poison the next call (vanilla.go:164-167).  The reference
 * returns NO estimate for such an Update ((nil, err)): the other members of a failed filter's slot (pred_covariance, gain,
 * innovation, measurement) are unspecified for that step -- some kernels write them as they are formed, before the step is
 * known to succeed (SquareRoot / Information: I-, yhat; the split-lane kernels for 8 < n <= 16: all four) -- and the
 * single-filter host mirrors (Go, C++) turn a set status word into the reference's (nil, err); batch callers check status. */
typedef struct kb_estimate_view {
    double *state;           /* [count][n]                                                          */
    double *covariance;      /* [count][n][n]                                                       */
    double *pred_covariance; /* [count][n][n]                                                       */
    double *gain;            /* [count][n][p]                                                       */
    double *innovation;      /* [count][p]   (KB_INFORMATION / KB_SRIF: [count][n], the information vector) */
    double *measurement;     /* [count][p]                                                          */
    uint32_t *status;        /* [count]                                                             */
    int clear_status;
} kb_estimate_view;
int kb_get_estimate(kb_batch *b, int64_t first, int64_t count, kb_estimate_view *view);
/* The reference's `est, err := kf.Update(y, u)` in ONE call and ONE synchronisation: the step (kb_update / kb_update_nl /
 * kb_predict_nl) and the snapshot of the estimate it produced are enqueued back to back.  This is what the host shims call for
 * every Update of a drop-in filter (one filter: 22 us per step against 32 us for kb_update + kb_get_estimate). */
int kb_update_estimate(kb_batch *b, const double *meas, int meas_rows, const double *ctrl, int ctrl_rows, int64_t first,
                       int64_t count, kb_estimate_view *view);
int kb_update_nl_estimate(kb_batch *b, const double *real_obs, int real_rows, const double *computed_obs,
                          int computed_rows, int64_t first, int64_t count, kb_estimate_view *view);
int kb_predict_nl_estimate(kb_batch *b, int64_t first, int64_t count, kb_estimate_view *view);
/* Planar device-side variant: writes element e of filter i to dst[e*ld + i] in the batch dtype. */
int kb_get_dev(kb_batch *b, int field, void *dst, int64_t ld);
int kb_get_status(kb_batch *b, uint32_t *host, int64_t first, int64_t count);
int kb_clear_status(kb_batch *b);
/* Estimate.IsWithinNsigma(N) (vanilla.go:231-239): out[i] = 1/0. */
int kb_is_within_nsigma(kb_batch *b, double nsigma, uint8_t *host, int64_t first, int64_t count);
/* kf.step.  The reference does not advance it on a failed Update (it returns before `kf.step++`: vanilla.go:164-167 / :207-215
 * against :218; srif.go:112-114; hybrid.go:150-152), so a filter's counter -- the index of its BatchNoise vectors, the k of its
 * error messages -- falls behind the number of calls by the steps that failed FOR THAT FILTER.  The engine keeps that count per
 * filter on the device.  kb_step: for a batch of at most 64 filters (the drop-in use: 1) the exact kf.step of filter 0, without
 * a device read (valid after a synchronising call); for larger batches the counter of a filter that never failed.
 * kb_filter_step: the exact kf.step of any filter (synchronises).  kb_calls: step / reset calls accepted so far (monotone:
 * what a host mirror compares to tell a live estimate view from a stale one). */
int64_t kb_step(const kb_batch *b);
int kb_filter_step(kb_batch *b, int64_t filter, int64_t *step);
int64_t kb_calls(const kb_batch *b);
int kb_need_ctrl(const kb_batch *b);      /* kf.needCtrl (vanilla.go:39) */
int kb_meas_dim(const kb_batch *b);       /* current rows of H */
int64_t kb_num_filters(const kb_batch *b);
void *kb_stream(const kb_batch *b);       /* the hipStream_t work is enqueued on */
int kb_synchronize(kb_batch *b);

/* ---- noise (noise.go:13-164) -------------------------------------------------- */
typedef enum {
    KB_NOISE_NOISELESS = 0, /* Noiseless (noise.go:23-64): w = v = 0               */
    KB_NOISE_AWGN = 1,      /* AWGN (noise.go:109-164): w ~ N(0,Q), v ~ N(0,R), device Philox4x32-10 + Box-Muller, x = L z with L = chol */
    KB_NOISE_BATCH = 2      /* BatchNoise (noise.go:67-106): pre-recorded vectors, see kb_set_batch_noise */
} kb_noise_kind;
/* Selects the Noise implementation; AWGN fails with KB_ERR_NOT_PD when Q or R is
 * not positive definite (the reference panics, noise.go:148-156).  `seed` replaces
 * the reference's wall-clock seed; every kb_reset moves to a fresh sub-stream. */
int kb_set_noise_kind(kb_batch *b, int noise_kind, uint64_t seed);
/* BatchNoise{process, measurement} (noise.go:67-106): process[nproc][n] and measurement[nmeas][p]
 * are the recorded vectors (host, row-major), shared by every filter of the batch; step k adds
 * process[k] at both Process(k) call sites and measurement[k] to yhat.  An Update at a step with no
 * recorded vector fails with "no process noise defined at step k=%d" (a panic in the reference,
 * noise.go:75-86).  BatchNoise reports zero Q and R (noise.go:89-98): this call zeroes the batch's Q and R.
 * Selects KB_NOISE_BATCH. */
int kb_set_batch_noise(kb_batch *b, const double *process, int nproc, const double *measurement, int nmeas);
/* The standard normals z behind the AWGN draw of the filter with global index `filter` at
 * (epoch, step, which); the noise vector is chol_L(Q) z (which 0, 2; n values) or
 * chol_L(R) z (which 1; p values).  For tests that replay the device's samples through the
 * oracle.  which: 0 = Process (first call), 1 = Measurement, 2 = Process (second call).  The values are the device's bit
 * for bit: Philox4x32-10 and a Box-Muller transform whose logarithm / sine / cosine (csrc/kb_normal.h) use only operations that
 * round identically on the host and on the GPU. */
int kb_noise_sample(kb_batch *b, int64_t filter, int64_t epoch, int64_t step, int which, double *out);
/* The same generator as a pure host function (no handle, no device): standard normal number k of
 * the vector drawn by global filter index `filter` at (epoch, step, which) under `seed`. */
double kb_noise_normal(uint64_t seed, int64_t filter, int64_t epoch, int64_t step, int which, int k);

/* ---- Monte-Carlo fan-out (montecarlo.go:92-119, 18-59) --------------------------- */
/* NewMonteCarloRuns(samples = N of the batch, steps, rowsH, controls, kf): `b` must be a
 * KB_VANILLA_PREDICT batch (the reference panics otherwise, montecarlo.go:93-95) with
 * AWGN noise.  controls: host [steps][m], or [1][m] meaning zero controls for every step
 * (montecarlo.go:98-104), anything else is an error (a panic there).  Runs all N runs x
 * `steps` steps in one launch and accumulates, per step and state component, over this
 * batch's runs:  sums[steps][3][n] (host, float64) = { sum(x - c), sum((x - c)^2), c }, where
 * c is the noise-free trajectory (identical on every shard; subtracting it before squaring
 * keeps the unbiased variance well conditioned).  Shards are combined by adding rows 0 and 1.
 * The batch is left Reset() (montecarlo.go:116).
 * first_run = global index of this batch's first run (sharding across GPUs: the noise
 * stream of a run depends only on its global index). */
int kb_mc_run(kb_batch *b, int steps, const double *controls, int ncontrols,
              int64_t first_run, double *sums);
/* Same with options.  KB_MC_KEEP_RUNS keeps what differs between the runs of MonteCarloRuns.Runs[r].Estimates[k]
 * (montecarlo.go:11-15, :108-117) on the device -- State() = x_k and Measurement() = yhat_k = H x_{k-1} + v_k of every run
 * and step, in the batch dtype, (n + p) x steps x N values -- for kb_mc_get_runs; AsCSV (montecarlo.go:62-89) and the
 * NewChiSquare of the reference read exactly these.  Covariance / PredCovariance / Gain of those estimates do not depend on
 * the run (a one-filter Noiseless batch stepped `steps` times gives them).  Refused above KB_MC_KEEP_MAX_BYTES: the
 * statistics (Mean / StdDev, kb_chisquare with replay_last_mc) never need the trajectories. */
#define KB_MC_KEEP_RUNS 0x1u
#define KB_MC_KEEP_MAX_BYTES (8ll << 30)
int kb_mc_run_ex(kb_batch *b, int steps, const double *controls, int ncontrols, int64_t first_run, double *sums,
                 unsigned mc_flags);
/* Runs [first, first + count) of the last kb_mc_run_ex(..., KB_MC_KEEP_RUNS): states[count][steps][n] and
 * measurements[count][steps][p] (host, float64; either may be NULL). */
int kb_mc_get_runs(kb_batch *b, int64_t first, int64_t count, double *states, double *measurements);
/* MonteCarloRuns.Mean / StdDev (montecarlo.go:18-59) from (all-reduced) sums over
 * `runs` runs: mean[steps][n], stddev[steps][n] (unbiased, n-1, as gonum stat.StdDev). */
int kb_mc_stats(const double *sums, int steps, int n, int64_t runs, double *mean, double *stddev);

/* ---- VanLoan continuous -> discrete conversion (c2d.go:13-75) ------------------------------ */
/* VanLoan(A, Gamma, W, dt) for N independent systems: A n x n, Gamma n x q, W q x q (row-major), dt scalar
 *   -> F = exp(A dt) n x n and Q n x n (symmetric: upper triangle mirrored, as AsSymDense returns it),
 * computed from exp([[-A dt, Gamma W Gamma^T dt], [0, A^T dt]]) like the reference.  n <= 8.
 * Host arrays hold N consecutive matrices, or one when the matching `broadcast` bit is set
 * (1 = A, 2 = Gamma, 4 = W, 8 = dt).  status[i] (may be NULL): KB_ST_NYQUIST = the reference's error
 * value (F and Q are valid next to it, c2d.go:26-28,74), KB_ST_ASYMMETRIC = the reference's QSym would
 * be nil (c2d.go:73), KB_ST_SINGULAR / KB_ST_NONFINITE = the Pade system of the exponential broke down.
 * The arithmetic runs in `dtype`. */
int kb_van_loan(int device, int dtype, int n, int q, int64_t N, const double *A, const double *Gamma,
                const double *W, const double *dt, int broadcast, double *F, double *Q, uint32_t *status);
/* Same on HBM-resident planar arrays in `dtype` (element e of system i at ptr[e*ld + i], dt[i]); F and Q
 * come out in the layout kb_set_dev(KB_F / KB_Q) reads.  Asynchronous on `stream` (a hipStream_t, e.g.
 * kb_stream(b); NULL = the default stream) of the current device. */
int kb_van_loan_dev(int dtype, int n, int q, int64_t N, const void *A, const void *Gamma, const void *W,
                    const void *dt, int64_t ld, void *F, void *Q, uint32_t *status, void *stream);

/* ---- chi-square consistency tests (chisquare.go:16-95) ------------------------------------- */
/* NewChiSquare(kf, runs, controls, withNEES, withNIS) fused with the truth generation of
 * NewMonteCarloRuns: `truth` is the pure-predictor Vanilla batch with AWGN noise (one run per
 * filter), `kf` a Vanilla batch of the same size and shape holding the filter under test (it is
 * Reset() for every run, chisquare.go:39, i.e. started from its initial estimate; neither batch's
 * current estimate is modified).  Per step and run: the truth advances, the filter is updated with
 * the truth's measurement, NEES = (x - xhat)^T P^-1 (x - xhat) and NIS = innov^T (H P- H^T + R)^-1
 * innov.  sums[steps][2] (host) receives { sum NIS, sum NEES } over this batch's runs; the step
 * means are sums / total runs (after adding the shards of a multi-GPU job).
 * replay_last_mc != 0 re-uses the noise epoch of the last kb_mc_run on `truth`, so the statistics
 * refer to the same runs as its means; 0 draws fresh runs.  controls as in kb_mc_run. */
int kb_chisquare(kb_batch *truth, kb_batch *kf, int steps, const double *controls, int ncontrols,
                 int64_t first_run, int replay_last_mc, int with_nees, int with_nis, double *sums);

/* ---- one process, every GPU of the node ----------------------------------- */
/* A batch of N filters split into contiguous shards over `ndev` devices -- shard g owns the filters [g N / G, (g + 1) N / G)
 * -- with one kb_batch, one host thread and one HIP stream per shard.  Filters share nothing (vanilla.go:216-218), so the
 * update path has no collective; the one exchange, the Monte-Carlo / chi-square statistics (montecarlo.go:18-59,
 * chisquare.go:85-94), is ONE ncclAllReduce(sum) of steps x 2n (resp. steps x 2) doubles over RCCL / xGMI, from this single
 * process (ncclCommInitAll; librccl is loaded on first use).  Shards that share a device, or a box without librccl, add the
 * shards on the host instead (kb_sharded_used_rccl tells which).  devices = NULL means 0 .. ndev-1.  Every call below fans
 * out to the shards' threads and returns when all are done; kb_sharded_shard(s, g) hands out a shard's own handle for
 * everything not wrapped here (device-resident setters, getters, the NLDKF calls, ...). */
typedef struct kb_sharded kb_sharded;
int kb_sharded_create(kb_sharded **out, int kind, int n, int p, int m, int64_t nfilters, int dtype, const int *devices,
                      int ndev, unsigned flags);
void kb_sharded_destroy(kb_sharded *s);
int kb_sharded_num_shards(const kb_sharded *s);
kb_batch *kb_sharded_shard(kb_sharded *s, int g);
int64_t kb_sharded_first(const kb_sharded *s, int g); /* global index of shard g's first filter; g = num_shards: N */
/* kb_set: a per-filter array [N][elems_per_filter] is cut at the shard boundaries, a shared one (broadcast) goes to all */
int kb_sharded_set(kb_sharded *s, int field, const double *host, int64_t count, int broadcast, int p_rows,
                   int64_t elems_per_filter);
int kb_sharded_set_noise_kind(kb_sharded *s, int noise_kind, uint64_t seed);
int kb_sharded_init(kb_sharded *s);
int kb_sharded_reset(kb_sharded *s);
int kb_sharded_synchronize(kb_sharded *s);
/* LDKF.Update on every shard in parallel: host measurements [N][meas_rows] (kb_update), or every shard's measurements
 * already in ITS device's memory, meas[g] planar with leading dimension ld_meas[g] (kb_update_dev; asynchronous) */
int kb_sharded_update(kb_sharded *s, const double *meas, int meas_rows, const double *ctrl, int ctrl_rows);
int kb_sharded_update_dev(kb_sharded *s, const void *const *meas, const int64_t *ld_meas, const void *const *ctrl,
                          const int64_t *ld_ctrl);
int kb_sharded_get(kb_sharded *s, int field, double *host, int64_t first, int64_t count, int64_t elems_per_filter);
int kb_sharded_get_status(kb_sharded *s, uint32_t *host, int64_t first, int64_t count);
/* NewMonteCarloRuns / NewChiSquare over the whole node: one ensemble, shard g runs the runs [first(g), first(g + 1)) (a run's
 * noise depends only on its global index); sums as kb_mc_run / kb_chisquare, over ALL runs. */
int kb_sharded_mc_run(kb_sharded *s, int steps, const double *controls, int ncontrols, double *sums, unsigned mc_flags);
int kb_sharded_chisquare(kb_sharded *truth, kb_sharded *kf, int steps, const double *controls, int ncontrols,
                         int replay_last_mc, int with_nees, int with_nis, double *sums);
int kb_sharded_used_rccl(const kb_sharded *s); /* 1: the last statistics reduction went through ncclAllReduce, 0: host sum */

#ifdef __cplusplus
}
#endif
#endif /* GOKALMAN_AMD_H */
