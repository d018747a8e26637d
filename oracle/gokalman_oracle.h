/*
 * gokalman_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, reference-order restatement of the gokalman predict/update hot
 * path (one filter at a time, float64, row-major), used ONLY as the checker:
 *   - tests/            (parity of the HIP path, golden-vector pinning)
 *   - __graft_entry__.smoke()
 *   - bench.py's `cpu_baseline` leg
 * Nothing under gokalman_amd/ may include, link or call this file; the product
 * library (libgokalman_amd.so) has no CPU path at all.
 *
 * The reference (ChristopherRabotin/gokalman, Go) cannot be built here: there
 * is no Go toolchain in the image, the repo has no go.mod / vendor dir, and its
 * arithmetic lives in the un-vendored, un-pinned third-party modules
 * github.com/gonum/matrix/mat64, gonum/stat, gonum/stat/distmv, gonum/floats
 * (2016-era split repos, later merged into gonum.org/v1/gonum).  Their
 * published algorithms are restated here (LAPACK dgetf2/dgetri, dpotf2,
 * dgeqr2/dlarfg, floats.EqualWithinAbsOrRel, stat.StdDev n-1).
 *
 * PARITY PINS (tests/test_oracle_golden.py):
 *   - examples/jerkcar/{vanilla,information,sqrt}.csv   -> Vanilla, Information,
 *     SquareRoot, 2000 steps each, to the %f print precision (5.1e-7 abs)
 *   - helper_test.go:108-117  HouseholderTransf 3x3 KAT   (1e-15)
 *   - srif_test.go:31-56      measurementSRIFUpdate KAT   (1e-4)
 *   - srif_test.go:15-29      SRIF initial covariance round trip (1e-12)
 * PARITY UNPINNED (no fixture in the reference that runs without the external
 * `smd` propagator): SRIF.fullUpdate time update + whitening (srif.go:111-148),
 * HybridKF.fullUpdate (hybrid.go:104-204; cross-checked against the Vanilla
 * restatement, same algebra), AWGN sample values (wall-clock seed).
 */
#ifndef GOKALMAN_ORACLE_H
#define GOKALMAN_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXN 16            /* max state / measurement / control dimension   */
#define ORC_MAXD 32            /* max panel dimension (2n or n+p)               */

/* filter kinds (kalman.go / constructors) */
enum {
    ORC_VANILLA = 1,           /* vanilla.go:21   NewVanilla                    */
    ORC_VANILLA_PREDICT = 2,   /* vanilla.go:43   NewPurePredictorVanilla       */
    ORC_SQUAREROOT = 3,        /* squareroot.go:21                              */
    ORC_INFORMATION = 4,       /* information.go:20,65                          */
    ORC_SRIF = 5,              /* srif.go:14                                    */
    ORC_HYBRID = 6,            /* hybrid.go:23                                  */
    ORC_BATCH_LS = 7           /* batch.go:34  NewBatchKF (batch least squares)  */
};

/* return codes of orc_update & friends (the reference's `error` / panic sites) */
enum {
    ORC_OK = 0,
    ORC_ERR_SINGULAR = 1,      /* "could not invert `H*P_kp1_minus*H' + R`" etc. */
    ORC_ERR_ASYMMETRIC = 2,    /* AsSymDense failure (helper.go:75)              */
    ORC_ERR_LOCKED = 3,        /* "kf is locked (call Prepare() first)"          */
    ORC_ERR_DIMS = 4,
    ORC_ERR_NOTPD = 5          /* Cholesky of a non positive definite matrix     */
};

/* what orc_get returns */
enum {
    ORC_GET_STATE = 0,         /* Estimate.State()                              */
    ORC_GET_COVAR = 1,         /* Estimate.Covariance()      (n x n, full)      */
    ORC_GET_PRED_COVAR = 2,    /* Estimate.PredCovariance()  (n x n, full)      */
    ORC_GET_GAIN = 3,          /* Gain()                     (n x p)            */
    ORC_GET_INNOV = 4,         /* Estimate.Innovation()                         */
    ORC_GET_MEAS = 5,          /* Estimate.Measurement()                        */
    ORC_GET_RAW_VEC = 6,       /* i (Information), b (SRIF), x otherwise        */
    ORC_GET_RAW_MAT = 7,       /* S (SquareRoot), I (Information), R (SRIF), P  */
    ORC_GET_RAW_PRED_MAT = 8   /* S- / I- / Rbar / P-                           */
};

typedef struct orc_filter orc_filter;

/* --- gonum-like primitives (exported for the KAT tests) ------------------- */
int  orc_inverse(int n, const double *A, double *Ainv, double *cond);
int  orc_cholesky_lower(int n, const double *A, double *L);
void orc_qr_r(int m, int n, const double *A, double *R);
int  orc_as_sym_dense(int n, const double *M, double *S);
double orc_sign(double v);
void orc_householder_transf(double *A, int n, int m);          /* helper.go:142 */
int  orc_measurement_srif_update(int n, int m, const double *R, const double *H,
                                 const double *b, const double *y,
                                 double *Rk, double *bk, double *ek); /* srif.go:298 */

/* --- filter objects -------------------------------------------------------- */
/* LDKF kinds: vanilla / predict-only vanilla / squareroot / information.
 * For ORC_INFORMATION (x0,P0) are (i0,I0) as in NewInformation; use
 * orc_information_from_state for NewInformationFromState. */
orc_filter *orc_new_ldkf(int kind, int n, int p, int m,
                         const double *x0, const double *P0,
                         const double *F, const double *G, const double *H,
                         const double *Q, const double *R);
orc_filter *orc_information_from_state(int n, int p, int m,
                         const double *x0, const double *P0,
                         const double *F, const double *G, const double *H,
                         const double *Q, const double *R);
/* NLDKF kinds */
orc_filter *orc_new_srif(int n, int p, const double *x0, const double *P0,
                         const double *R, int non_tri_r);
orc_filter *orc_new_hybrid(int n, int p, const double *x0, const double *P0,
                           int nq, const double *Q, const double *R);
/* NewBatchKF (batch.go:34-38): normal-equation accumulator; orc_update_nl = SetNextMeasurement
 * (:41-61) with the H given to orc_prepare, ORC_GET_STATE / ORC_GET_COVAR = Solve() (:64-79). */
orc_filter *orc_new_batch_ls(int n, int p, const double *R);
void orc_free(orc_filter *f);

/* LDKF setters (Set* in vanilla.go:96-118, squareroot.go:85-114, information.go:117-138) */
void orc_set_state_transition(orc_filter *f, const double *F);
void orc_set_input_control(orc_filter *f, int m, const double *G);
void orc_set_measurement_matrix(orc_filter *f, int p, const double *H);
int  orc_set_noise(orc_filter *f, int p, const double *Q, const double *R);
void orc_reset(orc_filter *f);

/* LDKF.Update(measurement, control).  w_pred / v_meas / w_post are the three
 * Noise draws of one step in call order (vanilla.go:146,157,195); NULL = zero
 * (Noiseless). */
int orc_update(orc_filter *f, const double *y, const double *u,
               const double *w_pred, const double *v_meas, const double *w_post);

/* NLDKF */
void orc_prepare(orc_filter *f, const double *Phi, const double *Htilde);
void orc_prepare_pnt(orc_filter *f, const double *Gamma);
void orc_enable_ekf(orc_filter *f, int on);
int  orc_update_nl(orc_filter *f, const double *real_obs, const double *computed_obs);
int  orc_predict_nl(orc_filter *f);

int  orc_get(orc_filter *f, int what, double *out);
int  orc_step(const orc_filter *f);
int  orc_is_within_nsigma(orc_filter *f, double N);

/* --- batch driver (bench.py cpu_baseline; OpenMP over filters) ------------- */
/* N independent Vanilla filters, T steps, AoS row-major inputs:
 * x[N][n] P[N][n*n] F[N][n*n] H[N][p*n] Q[N][n*n] R[N][p*p] y[T][N][p].
 * x and P are updated in place.  Returns the number of filters that errored. */
long orc_vanilla_batch(long N, int T, int n, int p,
                       double *x, double *P, const double *F, const double *H,
                       const double *Q, const double *R, const double *y,
                       int threads);
/* same for the other LDKF kinds (kind = ORC_SQUAREROOT / ORC_INFORMATION);
 * x,P are (x,P) on input and output (covariance form), the filter's internal
 * form is built by the constructor as in the reference.  y holds `ypool` steps,
 * step k uses y[k % ypool]. */
long orc_ldkf_batch(int kind, long N, int T, int n, int p,
                    double *x, double *P, const double *F, const double *H,
                    const double *Q, const double *R, const double *y, int ypool,
                    int threads);
int orc_max_threads(void);

/* --- SmoothAll (hybrid.go:209-238, srif.go:165-192) ------------------------- */
int orc_smooth_all(int n, int steps, const double *Phi, double *x, double *P);

/* --- Monte-Carlo statistics (montecarlo.go:18-59) -------------------------- */
/* states[runs][n] for ONE step -> mean[n], stddev[n] (unbiased, n-1). */
void orc_mc_mean_stddev(long runs, int n, const double *states,
                        double *mean, double *stddev);

/* --- VanLoan (c2d.go:13-75), vanloan_oracle.c --------------------------------- */
int orc_expm(int n, const double *A, double *E);                       /* mat64.Dense.Exp */
int orc_eigvals(int n, const double *A, double *wr, double *wi);       /* mat64.Eigen values */
/* returns bit 0 = Nyquist error, bit 1 = Q asymmetric (QSym nil) */
int orc_van_loan(int n, int q, const double *A, const double *Gamma, const double *W, double dt,
                 double *F, double *Q);

#ifdef __cplusplus
}
#endif
#endif
