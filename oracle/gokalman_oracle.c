/*
 * gokalman_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see the header).
 *
 * Reference-order restatement of gokalman's predict/update hot path.  Every
 * function cites the reference file:line it follows (paths relative to the
 * reference checkout).  Bug-compatible on purpose: see the QUIRK notes.
 *
 * gonum (not on disk, unpinned) primitives are restated from their published
 * LAPACK-equivalent algorithms:
 *   mat64.Dense.Inverse  = Dgetrf(Dgetf2, partial pivoting) + Dgetri, with a
 *                          Condition error when cond_inf > 1e16 (result still
 *                          written) and Condition(+Inf) when exactly singular
 *   mat64.Cholesky       = Dpotrf (reads the upper triangle), L = U^T
 *   mat64.QR             = Dgeqrf -> Dgeqr2/Dlarfg/Dlarf  (R_kk = -sign(a_kk)|a_k|,
 *                          and NO reflection when the sub-column is zero or for
 *                          the last column of a square panel)
 *   mat64.SymDense.At    = reads the upper triangle only
 *   floats.EqualWithinAbsOrRel, stat.Mean, stat.StdDev (unbiased)
 */
#include "gokalman_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NN (ORC_MAXN * ORC_MAXN)
#define DD (ORC_MAXD * ORC_MAXD)

/* ------------------------------------------------------------------------ */
/* dense helpers, row-major                                                  */
/* ------------------------------------------------------------------------ */
/* C(r x c) = A(r x k) * B(k x c)   (gonum Dgemm, k ascending per element) */
static void mm(int r, int k, int c, const double *A, const double *B, double *C) {
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) {
            double s = 0.0;
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * c + j];
            C[i * c + j] = s;
        }
}
/* C(r x c) = A(r x k) * B(c x k)^T */
static void mm_nt(int r, int k, int c, const double *A, const double *B, double *C) {
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) {
            double s = 0.0;
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[j * k + l];
            C[i * c + j] = s;
        }
}
/* C(r x c) = A(k x r)^T * B(k x c) */
static void mm_tn(int r, int k, int c, const double *A, const double *B, double *C) {
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) {
            double s = 0.0;
            for (int l = 0; l < k; l++) s += A[l * r + i] * B[l * c + j];
            C[i * c + j] = s;
        }
}
/* C(r x c) = A(k x r)^T * B(c x k)^T */
static void mm_tt(int r, int k, int c, const double *A, const double *B, double *C) {
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) {
            double s = 0.0;
            for (int l = 0; l < k; l++) s += A[l * r + i] * B[j * k + l];
            C[i * c + j] = s;
        }
}
static void mv(int r, int c, const double *A, const double *x, double *y) {
    for (int i = 0; i < r; i++) {
        double s = 0.0;
        for (int j = 0; j < c; j++) s += A[i * c + j] * x[j];
        y[i] = s;
    }
}
static void mtv(int r, int c, const double *A, const double *x, double *y) { /* y = A^T x */
    for (int j = 0; j < c; j++) {
        double s = 0.0;
        for (int i = 0; i < r; i++) s += A[i * c + j] * x[i];
        y[j] = s;
    }
}
static void transpose(int r, int c, const double *A, double *At) {
    for (int i = 0; i < r; i++)
        for (int j = 0; j < c; j++) At[j * r + i] = A[i * c + j];
}
static int is_nil(int r, int c, const double *A) { /* helper.go:49-62 IsNil */
    if (!A) return 1;
    for (int i = 0; i < r * c; i++)
        if (A[i] != 0.0) return 0;
    return 1;
}
/* mat64.NewSymDense(n, vals): later reads go through the upper triangle. */
static void sym_from_upper(int n, const double *M, double *S) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) S[i * n + j] = (j >= i) ? M[i * n + j] : M[j * n + i];
}

/* ------------------------------------------------------------------------ */
/* gonum/floats                                                              */
/* ------------------------------------------------------------------------ */
static int eq_within_abs(double a, double b, double tol) { return a == b || fabs(a - b) <= tol; }
static int eq_within_rel(double a, double b, double tol) {
    if (a == b) return 1;
    double delta = fabs(a - b);
    if (delta <= DBL_MIN) return delta <= tol * DBL_MIN;
    return delta / fmax(fabs(a), fabs(b)) <= tol;
}
static int eq_within_abs_or_rel(double a, double b, double at, double rt) {
    return eq_within_abs(a, b, at) || eq_within_rel(a, b, rt);
}

/* helper.go:65-84 AsSymDense.  Returns ORC_OK or ORC_ERR_ASYMMETRIC.  The value
 * handed back is the UPPER triangle mirrored (NewSymDense), not (M+M^T)/2. */
int orc_as_sym_dense(int n, const double *M, double *S) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (i != j && !eq_within_abs_or_rel(M[j * n + i], M[i * n + j], 1e-6, 1e-2))
                return ORC_ERR_ASYMMETRIC;
    double tmp[DD];
    sym_from_upper(n, M, tmp);
    memcpy(S, tmp, sizeof(double) * n * n);
    return ORC_OK;
}

/* helper.go:133-138 Sign: +1 inside the 1e-12 dead band, else v/|v|. */
double orc_sign(double v) {
    if (eq_within_abs(v, 0.0, 1e-12)) return 1.0;
    return v / fabs(v);
}

/* ------------------------------------------------------------------------ */
/* mat64.Dense.Inverse                                                       */
/* ------------------------------------------------------------------------ */
/* returns 0 ok, 1 ill-conditioned (cond > 1e16, Ainv still written),
 * 2 exactly singular (Ainv holds the LU factors, as gonum leaves them). */
int orc_inverse(int n, const double *A, double *Ainv, double *cond_out) {
    double a[DD];
    int ipiv[ORC_MAXD];
    memcpy(a, A, sizeof(double) * n * n);
    double anorm = 0.0; /* infinity norm (gonum CondNorm = MaxRowSum) */
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += fabs(a[i * n + j]);
        if (s > anorm || s != s) anorm = s;
    }
    /* Dgetf2 */
    int singular = 0;
    for (int j = 0; j < n; j++) {
        int jp = j;
        double best = fabs(a[j * n + j]);
        for (int i = j + 1; i < n; i++)
            if (fabs(a[i * n + j]) > best) { best = fabs(a[i * n + j]); jp = i; }
        ipiv[j] = jp;
        if (a[jp * n + j] == 0.0) { singular = 1; continue; }
        if (jp != j)
            for (int c = 0; c < n; c++) { double t = a[j * n + c]; a[j * n + c] = a[jp * n + c]; a[jp * n + c] = t; }
        double piv = a[j * n + j];
        if (fabs(piv) >= DBL_MIN) {
            double r = 1.0 / piv;
            for (int i = j + 1; i < n; i++) a[i * n + j] *= r;
        } else {
            for (int i = j + 1; i < n; i++) a[i * n + j] /= piv;
        }
        for (int i = j + 1; i < n; i++) {
            double l = a[i * n + j];
            for (int c = j + 1; c < n; c++) a[i * n + c] -= l * a[j * n + c];
        }
    }
    if (singular) {
        memcpy(Ainv, a, sizeof(double) * n * n);
        if (cond_out) *cond_out = INFINITY;
        return 2;
    }
    /* Dgetri: Dtrti2 (upper, non-unit) */
    for (int j = 0; j < n; j++) {
        a[j * n + j] = 1.0 / a[j * n + j];
        double ajj = -a[j * n + j];
        /* x = U[0:j,0:j] * x, x = a[0:j, j]  (Dtrmv upper, notrans, non-unit) */
        for (int i = 0; i < j; i++) {
            double s = a[i * n + i] * a[i * n + j];
            for (int k = i + 1; k < j; k++) s += a[i * n + k] * a[k * n + j];
            a[i * n + j] = s;
        }
        for (int i = 0; i < j; i++) a[i * n + j] *= ajj;
    }
    /* solve inv(A)*L = inv(U) column by column from the right */
    double work[ORC_MAXD];
    for (int j = n - 1; j >= 0; j--) {
        for (int i = j + 1; i < n; i++) { work[i] = a[i * n + j]; a[i * n + j] = 0.0; }
        if (j < n - 1)
            for (int i = 0; i < n; i++) {
                double s = 0.0;
                for (int k = j + 1; k < n; k++) s += a[i * n + k] * work[k];
                a[i * n + j] -= s;
            }
    }
    for (int j = n - 2; j >= 0; j--) {
        int jp = ipiv[j];
        if (jp != j)
            for (int i = 0; i < n; i++) { double t = a[i * n + j]; a[i * n + j] = a[i * n + jp]; a[i * n + jp] = t; }
    }
    double inorm = 0.0;
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += fabs(a[i * n + j]);
        if (s > inorm || s != s) inorm = s;
    }
    double cond = anorm * inorm;
    memcpy(Ainv, a, sizeof(double) * n * n);
    if (cond_out) *cond_out = cond;
    if (!(cond <= 1e16)) return 1; /* matrix.ConditionTolerance; NaN counts as bad */
    return 0;
}

/* mat64.Cholesky.Factorize + LFromCholesky.  Reads the upper triangle. */
int orc_cholesky_lower(int n, const double *A, double *L) {
    double u[DD];
    memset(u, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; j++) {
        double ajj = A[j * n + j];
        for (int k = 0; k < j; k++) ajj -= u[k * n + j] * u[k * n + j];
        if (!(ajj > 0.0)) return ORC_ERR_NOTPD;
        ajj = sqrt(ajj);
        u[j * n + j] = ajj;
        for (int i = j + 1; i < n; i++) {
            double s = A[j * n + i];
            for (int k = 0; k < j; k++) s -= u[k * n + j] * u[k * n + i];
            u[j * n + i] = s / ajj;
        }
    }
    transpose(n, n, u, L);
    return ORC_OK;
}

/* mat64.QR.Factorize + RFromQR: Dgeqr2 with Dlarfg / Dlarf.  R is m x n with
 * zeros below the diagonal. */
void orc_qr_r(int m, int n, const double *A, double *R) {
    double a[DD];
    double w[ORC_MAXD];
    memcpy(a, A, sizeof(double) * m * n);
    int kmax = m < n ? m : n;
    for (int i = 0; i < kmax; i++) {
        /* Dlarfg(m-i, alpha=a[i][i], x=a[i+1:,i]) */
        double tau = 0.0;
        int len = m - i;
        if (len > 1) {
            double xnorm = 0.0;
            for (int r = i + 1; r < m; r++) xnorm += a[r * n + i] * a[r * n + i];
            xnorm = sqrt(xnorm);
            if (xnorm != 0.0) {
                double alpha = a[i * n + i];
                double beta = -copysign(hypot(alpha, xnorm), alpha);
                tau = (beta - alpha) / beta;
                double sc = 1.0 / (alpha - beta);
                for (int r = i + 1; r < m; r++) a[r * n + i] *= sc;
                a[i * n + i] = beta;
            }
        }
        if (i < n - 1 && tau != 0.0) {
            /* Dlarf left: C = (I - tau v v^T) C, v = [1; a[i+1:,i]], C = a[i:, i+1:] */
            for (int c = i + 1; c < n; c++) {
                double s = a[i * n + c];
                for (int r = i + 1; r < m; r++) s += a[r * n + i] * a[r * n + c];
                w[c] = s;
            }
            for (int c = i + 1; c < n; c++) {
                a[i * n + c] -= tau * w[c];
                for (int r = i + 1; r < m; r++) a[r * n + c] -= tau * a[r * n + i] * w[c];
            }
        }
    }
    for (int r = 0; r < m; r++)
        for (int c = 0; c < n; c++) R[r * n + c] = (c >= r) ? a[r * n + c] : 0.0;
}

/* helper.go:142-172 HouseholderTransf(A, n, m): A is (m+n) x (n+1), in place. */
void orc_householder_transf(double *A, int n, int m) {
    int rows = m + n, cols = n + 1;
    double u[ORC_MAXD];
    for (int k = 0; k < n; k++) {
        double sigma = 0.0;
        for (int i = k; i < rows; i++) sigma += pow(A[i * cols + k], 2);
        sigma = sqrt(sigma) * orc_sign(A[k * cols + k]);
        memset(u, 0, sizeof(u));
        u[k] = A[k * cols + k] + sigma;
        A[k * cols + k] = -sigma;
        for (int i = k + 1; i < rows; i++) u[i] = A[i * cols + k];
        double beta = 1.0 / (sigma * u[k]);
        for (int j = k + 1; j < n + 1; j++) {
            double gamma = 0.0;
            for (int i = k; i < rows; i++) gamma += u[i] * A[i * cols + j];
            gamma *= beta;
            for (int i = k; i < rows; i++) A[i * cols + j] = A[i * cols + j] - gamma * u[i];
            for (int i = k + 1; i < rows; i++) A[i * cols + k] = 0.0;
        }
    }
}

/* srif.go:298-340 measurementSRIFUpdate: A = [[R b],[H y]] -> Householder ->
 * Rk = A[:n,:n], bk = A[:n,n], ek = A[n:,n]. */
int orc_measurement_srif_update(int n, int m, const double *R, const double *H,
                                const double *b, const double *y,
                                double *Rk, double *bk, double *ek) {
    double A[DD + ORC_MAXD];
    int cols = n + 1;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) A[i * cols + j] = R[i * n + j];
        A[i * cols + n] = b[i];
    }
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < n; j++) A[(n + i) * cols + j] = H[i * n + j];
        A[(n + i) * cols + n] = y[i];
    }
    orc_householder_transf(A, n, m);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) Rk[i * n + j] = A[i * cols + j];
        bk[i] = A[i * cols + n];
    }
    for (int i = 0; i < m; i++) ek[i] = A[(n + i) * cols + n];
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* filter object                                                             */
/* ------------------------------------------------------------------------ */
struct orc_filter {
    int kind, n, p, m, step;
    int need_ctrl, predict_only;
    /* model */
    double F[NN], G[NN], H[NN], Q[NN], R[NN];
    /* SquareRoot: cached Cholesky factors (squareroot.go:100-114) */
    double sqrtQ[NN], sqrtR[NN];
    int sqrt_p; /* dimension of sqrtR */
    /* Information: cached inverses (information.go:39-50); rinv_p = dim of Rinv */
    double Finv[NN], Qinv[NN], Rinv[NN];
    int rinv_p;
    /* NLDKF */
    double Phi[NN], Htilde[NN], Gamma[NN];
    int nq, have_gamma, ekf, locked, snc, non_tri_r;
    double sqrt_inv_noise[NN]; /* QUIRK srif.go:48: holds chol_L(R), not its inverse */
    /* current estimate (prevEst) and initial estimate (initEst) */
    double x[ORC_MAXN], M[NN], Mpred[NN];      /* vec: x | i | b ; mat: P | S | I | R */
    double x0[ORC_MAXN], M0[NN], Mpred0[NN];
    double K[NN], innov[ORC_MAXN], meas[ORC_MAXN], dobs[ORC_MAXN];
    int est_p;      /* rows of meas/innov of the current estimate */
    int have_gain;
};

static void save_init(orc_filter *f) {
    memcpy(f->x0, f->x, sizeof(f->x));
    memcpy(f->M0, f->M, sizeof(f->M));
    memcpy(f->Mpred0, f->Mpred, sizeof(f->Mpred));
}

/* squareroot.go:100-114 SetNoise: Cholesky of Q and R, cached. */
static int sqrt_set_noise(orc_filter *f, int p, const double *Q, const double *R) {
    int e1 = orc_cholesky_lower(f->n, Q, f->sqrtQ);
    int e2 = orc_cholesky_lower(p, R, f->sqrtR);
    f->sqrt_p = p;
    return e1 ? e1 : e2;
}

/* constructor body on caller-provided storage; returns 0 on success */
static int ldkf_init(orc_filter *f, int kind, int n, int p, int m,
                     const double *x0, const double *P0,
                     const double *F, const double *G, const double *H,
                     const double *Q, const double *R) {
    if (n < 1 || n > ORC_MAXN || p < 1 || p > ORC_MAXN || m < 0 || m > ORC_MAXN) return 1;
    f->kind = kind; f->n = n; f->p = p; f->m = m;
    f->step = 0; f->have_gain = 0; f->locked = 0; f->snc = 0; f->ekf = 0;
    memset(f->G, 0, sizeof(double) * n * (m > 0 ? m : 1));
    memset(f->innov, 0, sizeof(f->innov));
    memset(f->meas, 0, sizeof(f->meas));
    memset(f->Mpred, 0, sizeof(double) * n * n);
    memcpy(f->F, F, sizeof(double) * n * n);
    if (G && m > 0) memcpy(f->G, G, sizeof(double) * n * m);
    memcpy(f->H, H, sizeof(double) * p * n);
    sym_from_upper(n, Q, f->Q);
    sym_from_upper(p, R, f->R);
    f->need_ctrl = !is_nil(n, m, (G && m > 0) ? G : NULL); /* vanilla.go:39 !IsNil(G) */
    f->predict_only = (kind == ORC_VANILLA_PREDICT);
    memcpy(f->x, x0, sizeof(double) * n);
    f->est_p = p;
    switch (kind) {
    case ORC_VANILLA:
    case ORC_VANILLA_PREDICT:
        sym_from_upper(n, P0, f->M); /* Covar0 is a mat64.Symmetric */
        break;
    case ORC_SQUAREROOT: { /* squareroot.go:33-49 */
        double P0s[NN];
        sym_from_upper(n, P0, P0s);
        if (orc_cholesky_lower(n, P0s, f->M) != ORC_OK) return 1;
        if (sqrt_set_noise(f, p, f->Q, f->R) != ORC_OK) return 1;
        break;
    }
    case ORC_INFORMATION: { /* information.go:20-53; x0,P0 are i0,I0 */
        sym_from_upper(n, P0, f->M);
        orc_inverse(n, f->F, f->Finv, NULL);   /* errors only printed */
        orc_inverse(n, f->Q, f->Qinv, NULL);
        orc_inverse(p, f->R, f->Rinv, NULL);
        f->rinv_p = p;
        break;
    }
    default:
        return 1;
    }
    save_init(f);
    return 0;
}

orc_filter *orc_new_ldkf(int kind, int n, int p, int m,
                         const double *x0, const double *P0,
                         const double *F, const double *G, const double *H,
                         const double *Q, const double *R) {
    orc_filter *f = (orc_filter *)calloc(1, sizeof(*f));
    if (ldkf_init(f, kind, n, p, m, x0, P0, F, G, H, Q, R) != 0) { free(f); return NULL; }
    return f;
}

/* information.go:65-81 NewInformationFromState (x0,P0 -> i0,I0) */
static int info_from_state(int n, const double *x0, const double *P0, double *i0, double *I0) {
    double P0s[NN], I0t[NN];
    sym_from_upper(n, P0, P0s);
    if (orc_inverse(n, P0s, I0t, NULL) != 0) {
        memset(I0, 0, sizeof(double) * n * n);
    } else if (orc_as_sym_dense(n, I0t, I0) != ORC_OK) {
        return 1; /* reference would carry a nil SymDense and crash */
    }
    mv(n, n, I0, x0, i0);
    return 0;
}

/* information.go:65-81 NewInformationFromState */
orc_filter *orc_information_from_state(int n, int p, int m,
                         const double *x0, const double *P0,
                         const double *F, const double *G, const double *H,
                         const double *Q, const double *R) {
    double I0[NN], i0[ORC_MAXN];
    if (info_from_state(n, x0, P0, i0, I0)) return NULL;
    return orc_new_ldkf(ORC_INFORMATION, n, p, m, i0, I0, F, G, H, Q, R);
}

/* srif.go:14-49 NewSRIF */
orc_filter *orc_new_srif(int n, int p, const double *x0, const double *P0,
                         const double *R, int non_tri_r) {
    if (n < 1 || n > ORC_MAXN || p < 1 || p > ORC_MAXN) return NULL;
    orc_filter *f = (orc_filter *)calloc(1, sizeof(*f));
    f->kind = ORC_SRIF; f->n = n; f->p = p; f->est_p = p;
    double I0[NN];
    memset(I0, 0, sizeof(I0));
    for (int i = 0; i < n; i++) I0[i * n + i] = 1.0 / P0[i * n + i]; /* assumes diagonal P0 */
    if (orc_cholesky_lower(n, I0, f->M) != ORC_OK) { free(f); return NULL; }
    mv(n, n, f->M, x0, f->x);                    /* b0 = R0*x0 */
    memcpy(f->Mpred, f->M, sizeof(f->M));        /* est0 = {.., R0, R0} */
    double Rs[NN], L[NN], Linv[NN];
    sym_from_upper(p, R, Rs);
    memcpy(f->R, Rs, sizeof(double) * p * p);
    if (orc_cholesky_lower(p, Rs, L) != ORC_OK) { free(f); return NULL; }
    if (orc_inverse(p, L, Linv, NULL) != 0) { free(f); return NULL; } /* srif.go:43-45 */
    /* QUIRK srif.go:48: the struct keeps &sqrtMeasNoise (L), not its inverse. */
    memcpy(f->sqrt_inv_noise, L, sizeof(double) * p * p);
    f->non_tri_r = non_tri_r;
    f->locked = 1;
    save_init(f);
    return f;
}

/* hybrid.go:23-34 NewHybridKF */
orc_filter *orc_new_hybrid(int n, int p, const double *x0, const double *P0,
                           int nq, const double *Q, const double *R) {
    if (n < 1 || n > ORC_MAXN || p < 1 || p > ORC_MAXN || nq < 0 || nq > ORC_MAXN) return NULL;
    orc_filter *f = (orc_filter *)calloc(1, sizeof(*f));
    f->kind = ORC_HYBRID; f->n = n; f->p = p; f->est_p = p; f->nq = nq;
    memcpy(f->x, x0, sizeof(double) * n);
    sym_from_upper(n, P0, f->M);
    if (Q && nq > 0) sym_from_upper(nq, Q, f->Q);
    sym_from_upper(p, R, f->R);
    f->locked = 1;
    save_init(f);
    return f;
}

/* batch.go:34-38 NewBatchKF */
orc_filter *orc_new_batch_ls(int n, int p, const double *R) {
    if (n < 1 || n > ORC_MAXN || p < 1 || p > ORC_MAXN) return NULL;
    orc_filter *f = (orc_filter *)calloc(1, sizeof(*f));
    f->kind = ORC_BATCH_LS; f->n = n; f->p = p; f->est_p = p;
    sym_from_upper(p, R, f->R);
    f->locked = 1;
    return f;
}

void orc_free(orc_filter *f) { free(f); }
int orc_step(const orc_filter *f) { return f->step; }

/* vanilla.go:96-98 / squareroot.go:85-87 / information.go:117-123 */
void orc_set_state_transition(orc_filter *f, const double *F) {
    memcpy(f->F, F, sizeof(double) * f->n * f->n);
    if (f->kind == ORC_INFORMATION) orc_inverse(f->n, f->F, f->Finv, NULL);
}
/* vanilla.go:101-103: needCtrl is NOT recomputed by the setter. */
void orc_set_input_control(orc_filter *f, int m, const double *G) {
    f->m = m;
    memset(f->G, 0, sizeof(f->G));
    if (G && m > 0) memcpy(f->G, G, sizeof(double) * f->n * m);
}
void orc_set_measurement_matrix(orc_filter *f, int p, const double *H) {
    f->p = p;
    memcpy(f->H, H, sizeof(double) * p * f->n);
}
/* vanilla.go:111-113, squareroot.go:100-114, information.go:136-138.
 * QUIRK: Information.SetNoise does not refresh Qinv/Rinv. */
int orc_set_noise(orc_filter *f, int p, const double *Q, const double *R) {
    sym_from_upper(f->n, Q, f->Q);
    sym_from_upper(p, R, f->R);
    if (f->kind == ORC_SQUAREROOT) return sqrt_set_noise(f, p, f->Q, f->R);
    return ORC_OK;
}
/* vanilla.go:121-125 Reset */
void orc_reset(orc_filter *f) {
    memcpy(f->x, f->x0, sizeof(f->x));
    memcpy(f->M, f->M0, sizeof(f->M));
    memcpy(f->Mpred, f->Mpred0, sizeof(f->Mpred));
    memset(f->innov, 0, sizeof(f->innov));
    memset(f->meas, 0, sizeof(f->meas));
    f->have_gain = 0;
    f->step = 0;
}

static void add_opt(int n, double *x, const double *w) {
    if (w) for (int i = 0; i < n; i++) x[i] += w[i];
}

/* ------------------------------------------------------------------------ */
/* vanilla.go:128-220 Vanilla.Update                                         */
/* ------------------------------------------------------------------------ */
static int vanilla_update(orc_filter *f, const double *y, const double *u,
                          const double *w1, const double *v, const double *w2) {
    const int n = f->n, p = f->p, m = f->m;
    double xm[ORC_MAXN], t[ORC_MAXN];
    /* :138-146 prediction */
    mv(n, n, f->F, f->x, xm);
    if (f->need_ctrl) {
        mv(n, m, f->G, u, t);
        for (int i = 0; i < n; i++) xm[i] = xm[i] + t[i];
    }
    add_opt(n, xm, w1);
    /* :149-152 P- = F P F^T + Q */
    double FP[NN], Pm[NN];
    mm(n, n, n, f->F, f->M, FP);
    mm_nt(n, n, n, FP, f->F, Pm);
    for (int i = 0; i < n * n; i++) Pm[i] += f->Q[i];
    /* :155-157 yhat = H x_prev + v   (previous posterior, not x-) */
    double yhat[ORC_MAXN];
    mv(p, n, f->H, f->x, yhat);
    add_opt(p, yhat, v);
    /* :160-168 gain */
    double PHt[NN], S[NN], Sinv[NN], K[NN];
    mm_nt(n, n, p, Pm, f->H, PHt);
    mm(p, n, p, f->H, PHt, S);
    for (int i = 0; i < p * p; i++) S[i] += f->R[i];
    if (orc_inverse(p, S, Sinv, NULL) != 0) return ORC_ERR_SINGULAR;
    mm(n, p, p, PHt, Sinv, K);

    if (f->predict_only) { /* :170-179 (AsSymDense error ignored) */
        double Ps[NN];
        sym_from_upper(n, Pm, Ps);
        memcpy(f->x, xm, sizeof(double) * n);
        memcpy(f->M, Ps, sizeof(double) * n * n);
        memcpy(f->Mpred, Ps, sizeof(double) * n * n);
        memcpy(f->meas, yhat, sizeof(double) * p);
        memset(f->innov, 0, sizeof(f->innov));
        memcpy(f->K, K, sizeof(double) * n * p);
        f->have_gain = 1; f->est_p = p;
        f->step++;
        return ORC_OK;
    }
    /* :182-195 measurement update */
    double innov[ORC_MAXN], xp[ORC_MAXN];
    mv(p, n, f->H, xm, t);
    for (int i = 0; i < p; i++) innov[i] = y[i] - t[i];
    if (p == 1) {
        for (int i = 0; i < n; i++) t[i] = innov[0] * K[i] + 0.0;
    } else {
        mv(n, p, K, innov, t);
    }
    for (int i = 0; i < n; i++) xp[i] = xm[i] + t[i];
    add_opt(n, xp, w2);
    /* :197-205 Joseph form */
    double A[NN], P1[NN], Pp[NN], KR[NN], KRKt[NN];
    mm(n, p, n, K, f->H, A);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[i * n + j] = ((i == j) ? 1.0 : 0.0) - A[i * n + j];
    mm(n, n, n, A, Pm, P1);
    mm_nt(n, n, n, P1, A, Pp);
    mm(n, p, p, K, f->R, KR);
    mm_nt(n, p, n, KR, K, KRKt);
    for (int i = 0; i < n * n; i++) Pp[i] += KRKt[i];
    /* :207-215 */
    double Pms[NN], Pps[NN];
    if (orc_as_sym_dense(n, Pm, Pms) != ORC_OK) return ORC_ERR_ASYMMETRIC;
    if (orc_as_sym_dense(n, Pp, Pps) != ORC_OK) return ORC_ERR_ASYMMETRIC;
    memcpy(f->x, xp, sizeof(double) * n);
    memcpy(f->M, Pps, sizeof(double) * n * n);
    memcpy(f->Mpred, Pms, sizeof(double) * n * n);
    memcpy(f->meas, yhat, sizeof(double) * p);
    memcpy(f->innov, innov, sizeof(double) * p);
    memcpy(f->K, K, sizeof(double) * n * p);
    f->have_gain = 1; f->est_p = p;
    f->step++;
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* squareroot.go:129-274 SquareRoot.Update                                   */
/* ------------------------------------------------------------------------ */
static int squareroot_update(orc_filter *f, const double *y, const double *u,
                             const double *v, const double *w) {
    const int n = f->n, p = f->p, m = f->m;
    double xm[ORC_MAXN], t[ORC_MAXN];
    /* :139-147 (no process noise in the prediction) */
    mv(n, n, f->F, f->x, xm);
    if (f->need_ctrl) {
        mv(n, m, f->G, u, t);
        for (int i = 0; i < n; i++) xm[i] = xm[i] + t[i];
    }
    /* :155-175 C = [S^T F^T ; sqrtQ^T]  (2n x n) */
    double C[DD], Uc[DD];
    mm_tt(n, n, n, f->M, f->F, C);
    transpose(n, n, f->sqrtQ, C + n * n);
    /* :176-185 Uc = R factor; QUIRK: S- := Uc (upper), not Uc^T */
    orc_qr_r(2 * n, n, C, Uc);
    double Sm[NN];
    memcpy(Sm, Uc, sizeof(double) * n * n); /* top n x n block */
    /* :190-216 Delta = [[sqrtR^T, 0],[S-^T H^T, S-^T]] */
    double SmtHt[NN];
    mm_tt(n, n, p, Sm, f->H, SmtHt); /* n x p */
    const int d = n + p;
    const int sp = f->sqrt_p;
    double D[DD], UD[DD];
    for (int c = 0; c < d; c++)
        for (int r = 0; r < d; r++) {
            double val;
            if (c < sp) {
                if (r < sp) val = f->sqrtR[c * sp + r];          /* sqrtR^T[r][c] */
                else        val = SmtHt[(r - sp) * p + c];
            } else if (r < sp) {
                val = 0.0;
            } else {
                val = Sm[(c - p) * n + (r - sp)];                  /* S-^T[r-p][c-p] */
            }
            D[r * d + c] = val;
        }
    orc_qr_r(d, d, D, UD);
    /* :225-234 */
    double Sp[NN], Syy[NN], W[NN];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Sp[i * n + j] = UD[(p + j) * d + (p + i)];
    for (int i = 0; i < p; i++)
        for (int j = 0; j < p; j++) Syy[i * p + j] = UD[j * d + i];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < p; j++) W[i * p + j] = UD[j * d + (p + i)];
    /* :237-239 */
    double yhat[ORC_MAXN];
    mv(p, n, f->H, f->x, yhat);
    add_opt(p, yhat, v);
    /* :242-252 gain; the inverse's error is never looked at (err vs invErr typo) */
    double SyyInv[NN], K[NN];
    orc_inverse(p, Syy, SyyInv, NULL);
    if (p == 1) {
        for (int i = 0; i < n; i++) K[i] = SyyInv[0] * W[i];
    } else {
        mm(n, p, p, W, SyyInv, K);
    }
    /* :255-268 */
    double innov[ORC_MAXN], xp[ORC_MAXN];
    mv(p, n, f->H, xm, t);
    for (int i = 0; i < p; i++) innov[i] = y[i] - t[i];
    if (p == 1) {
        for (int i = 0; i < n; i++) t[i] = innov[0] * K[i] + 0.0;
    } else {
        mv(n, p, K, innov, t);
    }
    for (int i = 0; i < n; i++) xp[i] = xm[i] + t[i];
    add_opt(n, xp, w);
    memcpy(f->x, xp, sizeof(double) * n);
    memcpy(f->M, Sp, sizeof(double) * n * n);
    memcpy(f->Mpred, Sm, sizeof(double) * n * n);
    memcpy(f->meas, yhat, sizeof(double) * p);
    memcpy(f->innov, innov, sizeof(double) * p);
    memcpy(f->K, K, sizeof(double) * n * p);
    f->have_gain = 1; f->est_p = p;
    f->step++;
    return ORC_OK;
}

/* information.go:257-293: State() = Covariance()*i, Covariance() = inverse(I)
 * or the zero matrix when Inverse reports a Condition error. */
static int info_covariance(int n, const double *I, double *P) {
    double t[NN];
    if (orc_inverse(n, I, t, NULL) != 0) {
        memset(P, 0, sizeof(double) * n * n);
        return 1;
    }
    if (orc_as_sym_dense(n, t, P) != ORC_OK) { /* reference: nil SymDense */
        memset(P, 0, sizeof(double) * n * n);
        return 2;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* information.go:153-227 Information.Update                                 */
/* ------------------------------------------------------------------------ */
static int information_update(orc_filter *f, const double *y, const double *u, const double *v) {
    const int n = f->n, p = f->p, m = f->m;
    /* :163-165 zk = Finv^T (I Finv) */
    double t1[NN], zk[NN];
    mm(n, n, n, f->M, f->Finv, t1);
    mm_tn(n, n, n, f->Finv, t1, zk);
    /* :169-174 zkzkqi = -zk (zk + Qinv)^-1   (inverse error ignored) */
    double zq[NN], zqi[NN], Z[NN];
    for (int i = 0; i < n * n; i++) zq[i] = zk[i] + f->Qinv[i];
    orc_inverse(n, zq, zqi, NULL);
    mm(n, n, n, zk, zqi, Z);
    for (int i = 0; i < n * n; i++) Z[i] = -1.0 * Z[i];
    /* :176-185 */
    double im[ORC_MAXN], t[ORC_MAXN], t2[ORC_MAXN];
    mtv(n, n, f->Finv, f->x, im);
    if (f->need_ctrl) {
        mv(n, m, f->G, u, t);
        mv(n, n, zk, t, t2);
        for (int i = 0; i < n; i++) im[i] = im[i] + t2[i];
    }
    double IZ[NN];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) IZ[i * n + j] = ((i == j) ? 1.0 : 0.0) + Z[i * n + j];
    mv(n, n, IZ, im, t);
    memcpy(im, t, sizeof(double) * n);
    /* :188-190 I- = zk + Z zk^T */
    double Im[NN];
    mm_nt(n, n, n, Z, zk, Im);
    for (int i = 0; i < n * n; i++) Im[i] = zk[i] + Im[i];
    /* :192-194 yhat = H State(prev) + v */
    double Pprev[NN], xprev[ORC_MAXN], yhat[ORC_MAXN];
    info_covariance(n, f->M, Pprev);
    mv(n, n, Pprev, f->x, xprev);
    mv(p, n, f->H, xprev, yhat);
    add_opt(p, yhat, v);
    /* :197-203 HTR = H^T Rinv; QUIRK: a stale 1x1 Rinv acts as a scalar */
    double HTR[NN];
    if (f->rinv_p == 1) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < p; j++) HTR[i * p + j] = f->Rinv[0] * f->H[j * n + i];
    } else {
        if (f->rinv_p != p) return ORC_ERR_DIMS; /* mat64 would panic */
        mm_tn(n, p, p, f->H, f->Rinv, HTR);
    }
    /* :205-212 */
    double ip[ORC_MAXN], Ip[NN];
    mv(n, p, HTR, y, ip);
    for (int i = 0; i < n; i++) ip[i] = ip[i] + im[i];
    mm(n, p, n, HTR, f->H, Ip);
    for (int i = 0; i < n * n; i++) Ip[i] = Im[i] + Ip[i];
    /* :214-222 (the reference panics) */
    double Ims[NN], Ips[NN];
    if (orc_as_sym_dense(n, Im, Ims) != ORC_OK) return ORC_ERR_ASYMMETRIC;
    if (orc_as_sym_dense(n, Ip, Ips) != ORC_OK) return ORC_ERR_ASYMMETRIC;
    memcpy(f->x, ip, sizeof(double) * n);
    memcpy(f->M, Ips, sizeof(double) * n * n);
    memcpy(f->Mpred, Ims, sizeof(double) * n * n);
    memcpy(f->meas, yhat, sizeof(double) * p);
    f->est_p = p; f->have_gain = 0;
    f->step++;
    return ORC_OK;
}

int orc_update(orc_filter *f, const double *y, const double *u,
               const double *w_pred, const double *v_meas, const double *w_post) {
    switch (f->kind) {
    case ORC_VANILLA:
    case ORC_VANILLA_PREDICT: return vanilla_update(f, y, u, w_pred, v_meas, w_post);
    case ORC_SQUAREROOT:      return squareroot_update(f, y, u, v_meas, w_post);
    case ORC_INFORMATION:     return information_update(f, y, u, v_meas);
    default: return ORC_ERR_DIMS;
    }
}

/* ------------------------------------------------------------------------ */
/* NLDKF: srif.go:82-160, hybrid.go:78-204                                   */
/* ------------------------------------------------------------------------ */
void orc_prepare(orc_filter *f, const double *Phi, const double *Htilde) {
    memcpy(f->Phi, Phi, sizeof(double) * f->n * f->n);
    memcpy(f->Htilde, Htilde, sizeof(double) * f->p * f->n);
    f->locked = 0;
}
void orc_prepare_pnt(orc_filter *f, const double *Gamma) { /* hybrid.go:86-89 */
    if (f->kind != ORC_HYBRID) return;                      /* srif.go:79 no-op  */
    memcpy(f->Gamma, Gamma, sizeof(double) * f->n * f->nq);
    f->have_gamma = 1;
    f->snc = 1;
}
void orc_enable_ekf(orc_filter *f, int on) { if (f->kind == ORC_HYBRID) f->ekf = on ? 1 : 0; }

/* srif.go:223-234 SRIFEstimate.State(): x = R^-1 b (panics when singular) */
static int srif_state(int n, const double *R, const double *b, double *x) {
    double Ri[NN];
    if (orc_inverse(n, R, Ri, NULL) != 0) return ORC_ERR_SINGULAR;
    mv(n, n, Ri, b, x);
    return ORC_OK;
}

/* srif.go:101-160 */
static int srif_full_update(orc_filter *f, int pure, const double *real_obs, const double *computed) {
    const int n = f->n, p = f->p;
    if (f->locked) return ORC_ERR_LOCKED;
    double invPhi[NN], RBar[NN];
    if (orc_inverse(n, f->Phi, invPhi, NULL) != 0) return ORC_ERR_SINGULAR; /* :111-114 */
    mm(n, n, n, f->M, invPhi, RBar);                                        /* :115 */
    double xprev[ORC_MAXN], xBar[ORC_MAXN], bBar[ORC_MAXN];
    int e = srif_state(n, f->M, f->x, xprev);
    if (e) return e;
    mv(n, n, f->Phi, xprev, xBar);                                          /* :118 */
    mv(n, n, RBar, xBar, bBar);                                             /* :119 */
    /* :121-132 "make Rbar triangular": Augment then slice back out -- a copy;
     * HouseholderTransf is never called here (no-op quirk). */
    if (pure) { /* :134-141 */
        memcpy(f->x, bBar, sizeof(double) * n);
        memcpy(f->M, RBar, sizeof(double) * n * n);
        memcpy(f->Mpred, RBar, sizeof(double) * n * n);
        memset(f->meas, 0, sizeof(f->meas));
        memset(f->dobs, 0, sizeof(f->dobs));
        f->step++;
        f->locked = 1;
        return ORC_OK;
    }
    double yv[ORC_MAXN], yw[ORC_MAXN], Hw[NN];
    for (int i = 0; i < p; i++) yv[i] = real_obs[i] - computed[i];          /* :143-144 */
    mm(p, p, n, f->sqrt_inv_noise, f->Htilde, Hw);                          /* :146-147 */
    mv(p, p, f->sqrt_inv_noise, yv, yw);                                    /* :148 */
    double Rk[NN], bk[ORC_MAXN], ek[ORC_MAXN];
    orc_measurement_srif_update(n, p, RBar, Hw, bBar, yw, Rk, bk, ek);      /* :150 */
    memcpy(f->x, bk, sizeof(double) * n);
    memcpy(f->M, Rk, sizeof(double) * n * n);
    memcpy(f->Mpred, RBar, sizeof(double) * n * n);
    memcpy(f->meas, real_obs, sizeof(double) * p);
    memcpy(f->dobs, yw, sizeof(double) * p);
    f->step++;
    f->locked = 1;
    return ORC_OK;
}

/* hybrid.go:104-204 */
static int hybrid_full_update(orc_filter *f, int pure, const double *real_obs, const double *computed) {
    const int n = f->n, p = f->p, q = f->nq;
    if (f->locked) return ORC_ERR_LOCKED;
    double PhiP[NN], PBar[NN];
    mm(n, n, n, f->Phi, f->M, PhiP);                                        /* :114-116 */
    mm_nt(n, n, n, PhiP, f->Phi, PBar);
    if (f->snc) {                                                           /* :117-123 */
        double GQ[NN], GQGt[NN];
        mm(n, q, q, f->Gamma, f->Q, GQ);
        mm_nt(n, q, n, GQ, f->Gamma, GQGt);
        for (int i = 0; i < n * n; i++) PBar[i] += GQGt[i];
    }
    if (pure) {                                                             /* :125-143 */
        double xBar[ORC_MAXN], Ps[NN];
        if (f->ekf) memset(xBar, 0, sizeof(xBar));
        else mv(n, n, f->Phi, f->x, xBar);
        if (orc_as_sym_dense(n, PBar, Ps) != ORC_OK) return ORC_ERR_ASYMMETRIC;
        memcpy(f->x, xBar, sizeof(double) * n);
        memcpy(f->M, Ps, sizeof(double) * n * n);
        memcpy(f->Mpred, Ps, sizeof(double) * n * n);
        memset(f->meas, 0, sizeof(f->meas));
        memset(f->innov, 0, sizeof(f->innov));
        memset(f->dobs, 0, sizeof(f->dobs));
        memset(f->K, 0, sizeof(f->K));
        f->have_gain = 0;
        f->step++; f->snc = 0; f->locked = 1;
        return ORC_OK;
    }
    double PHt[NN], S[NN], Sinv[NN], K[NN];
    mm_nt(n, n, p, PBar, f->Htilde, PHt);                                   /* :146-153 */
    mm(p, n, p, f->Htilde, PHt, S);
    for (int i = 0; i < p * p; i++) S[i] += f->R[i];
    if (orc_inverse(p, S, Sinv, NULL) != 0) return ORC_ERR_SINGULAR;
    mm(n, p, p, PHt, Sinv, K);
    double yv[ORC_MAXN], innov[ORC_MAXN], xHat[ORC_MAXN], t[ORC_MAXN];
    for (int i = 0; i < p; i++) yv[i] = real_obs[i] - computed[i];          /* :156-157 */
    memset(innov, 0, sizeof(innov));
    if (f->ekf) {
        mv(n, p, K, yv, xHat);                                              /* :160-161 */
    } else {
        double xBar[ORC_MAXN];
        mv(n, n, f->Phi, f->x, xBar);                                       /* :164-165 */
        mv(p, n, f->Htilde, xBar, t);
        for (int i = 0; i < p; i++) innov[i] = yv[i] - t[i];
        mv(n, p, K, innov, t);
        for (int i = 0; i < n; i++) xHat[i] = xBar[i] + t[i];
    }
    double A[NN], P1[NN], Pp[NN], KR[NN], KRKt[NN];                         /* :174-182 */
    mm(n, p, n, K, f->Htilde, A);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[i * n + j] = ((i == j) ? 1.0 : 0.0) - A[i * n + j];
    mm(n, n, n, A, PBar, P1);
    mm_nt(n, n, n, P1, A, Pp);
    mm(n, p, p, K, f->R, KR);
    mm_nt(n, p, n, KR, K, KRKt);
    for (int i = 0; i < n * n; i++) Pp[i] += KRKt[i];
    double Pbs[NN], Pps[NN];
    if (orc_as_sym_dense(n, PBar, Pbs) != ORC_OK) return ORC_ERR_ASYMMETRIC;
    if (orc_as_sym_dense(n, Pp, Pps) != ORC_OK) return ORC_ERR_ASYMMETRIC;
    memcpy(f->x, xHat, sizeof(double) * n);
    memcpy(f->M, Pps, sizeof(double) * n * n);
    memcpy(f->Mpred, Pbs, sizeof(double) * n * n);
    memcpy(f->meas, real_obs, sizeof(double) * p);
    memcpy(f->innov, innov, sizeof(double) * p);
    memcpy(f->dobs, yv, sizeof(double) * p);
    memcpy(f->K, K, sizeof(double) * n * p);
    f->have_gain = 1;
    f->step++; f->snc = 0; f->locked = 1;
    return ORC_OK;
}

/* batch.go:41-61 SetNextMeasurement: Lambda += H^T R H, N += H^T R (real - computed).
 * QUIRK: the weight is the measurement noise matrix R itself, not its inverse. */
static int batch_ls_add(orc_filter *f, const double *real_obs, const double *computed) {
    const int n = f->n, p = f->p;
    double HtR[NN], HtRH[NN], y[ORC_MAXN], t[ORC_MAXN];
    mm_tn(n, p, p, f->Htilde, f->R, HtR);
    mm(n, p, n, HtR, f->Htilde, HtRH);
    for (int i = 0; i < n * n; i++) f->M[i] = f->M[i] + HtRH[i];
    for (int i = 0; i < p; i++) y[i] = real_obs[i] - computed[i];
    mv(n, p, HtR, y, t);
    for (int i = 0; i < n; i++) f->x[i] = f->x[i] + t[i];
    f->step++;
    return ORC_OK;
}

int orc_update_nl(orc_filter *f, const double *real_obs, const double *computed_obs) {
    if (f->kind == ORC_BATCH_LS) return batch_ls_add(f, real_obs, computed_obs);
    if (f->kind == ORC_SRIF) return srif_full_update(f, 0, real_obs, computed_obs);
    if (f->kind == ORC_HYBRID) return hybrid_full_update(f, 0, real_obs, computed_obs);
    return ORC_ERR_DIMS;
}
int orc_predict_nl(orc_filter *f) {
    if (f->kind == ORC_SRIF) return srif_full_update(f, 1, NULL, NULL);
    if (f->kind == ORC_HYBRID) return hybrid_full_update(f, 1, NULL, NULL);
    return ORC_ERR_DIMS;
}

/* ------------------------------------------------------------------------ */
/* Estimate getters                                                          */
/* ------------------------------------------------------------------------ */
/* srif.go:253-281: P = R^-1 R^-T, upper triangle mirrored */
static int srif_covariance(int n, const double *R, double *P) {
    double Ri[NN], t[NN];
    if (orc_inverse(n, R, Ri, NULL) != 0) { memset(P, 0, sizeof(double) * n * n); return 1; }
    mm_nt(n, n, n, Ri, Ri, t);
    sym_from_upper(n, t, P);
    return 0;
}

int orc_get(orc_filter *f, int what, double *out) {
    const int n = f->n, p = f->est_p;
    double t[NN];
    switch (what) {
    case ORC_GET_STATE:
        if (f->kind == ORC_BATCH_LS) { /* batch.go:64-79 Solve: P0 = AsSymDense(inverse(Lambda)), xHat0 = P0 N */
            double Li[NN], P0[NN];
            if (orc_inverse(n, f->M, Li, NULL) != 0) return ORC_ERR_SINGULAR;
            if (orc_as_sym_dense(n, Li, P0) != ORC_OK) return ORC_ERR_ASYMMETRIC;
            mv(n, n, P0, f->x, out);
            return ORC_OK;
        }
        if (f->kind == ORC_INFORMATION) {
            double P[NN];
            info_covariance(n, f->M, P);
            mv(n, n, P, f->x, out);
        } else if (f->kind == ORC_SRIF) {
            return srif_state(n, f->M, f->x, out);
        } else {
            memcpy(out, f->x, sizeof(double) * n);
        }
        return ORC_OK;
    case ORC_GET_COVAR:
    case ORC_GET_PRED_COVAR: {
        const double *Msrc = (what == ORC_GET_COVAR) ? f->M : f->Mpred;
        if (f->kind == ORC_BATCH_LS) {
            double Li[NN];
            if (orc_inverse(n, f->M, Li, NULL) != 0) return ORC_ERR_SINGULAR;
            return orc_as_sym_dense(n, Li, out);
        }
        if (f->kind == ORC_SQUAREROOT) {          /* squareroot.go:317-340 */
            mm_nt(n, n, n, Msrc, Msrc, t);
            sym_from_upper(n, t, out);
        } else if (f->kind == ORC_INFORMATION) {  /* information.go:277-316 */
            info_covariance(n, Msrc, out);
        } else if (f->kind == ORC_SRIF) {
            srif_covariance(n, Msrc, out);
        } else {
            memcpy(out, Msrc, sizeof(double) * n * n);
        }
        return ORC_OK;
    }
    case ORC_GET_GAIN:
        if (!f->have_gain) memset(out, 0, sizeof(double) * n * p);
        else memcpy(out, f->K, sizeof(double) * n * p);
        return ORC_OK;
    case ORC_GET_INNOV: /* information.go:272 / srif.go:237: returns the info vector */
        if (f->kind == ORC_INFORMATION || f->kind == ORC_SRIF) memcpy(out, f->x, sizeof(double) * n);
        else memcpy(out, f->innov, sizeof(double) * p);
        return ORC_OK;
    case ORC_GET_MEAS:
        memcpy(out, f->meas, sizeof(double) * p);
        return ORC_OK;
    case ORC_GET_RAW_VEC:
        memcpy(out, f->x, sizeof(double) * n);
        return ORC_OK;
    case ORC_GET_RAW_MAT:
        memcpy(out, f->M, sizeof(double) * n * n);
        return ORC_OK;
    case ORC_GET_RAW_PRED_MAT:
        memcpy(out, f->Mpred, sizeof(double) * n * n);
        return ORC_OK;
    }
    return ORC_ERR_DIMS;
}

/* vanilla.go:231-239 etc. IsWithinNsigma */
int orc_is_within_nsigma(orc_filter *f, double N) {
    double x[ORC_MAXN], P[NN];
    if (orc_get(f, ORC_GET_STATE, x) != ORC_OK) return 0;
    orc_get(f, ORC_GET_COVAR, P);
    for (int i = 0; i < f->n; i++) {
        double ns = N * sqrt(P[i * f->n + i]);
        if (x[i] > ns || x[i] < -ns) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------------ */
/* batch drivers (cpu_baseline)                                              */
/* ------------------------------------------------------------------------ */
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

long orc_ldkf_batch(int kind, long N, int T, int n, int p,
                    double *x, double *P, const double *F, const double *H,
                    const double *Q, const double *R, const double *y, int ypool,
                    int threads) {
    long nerr = 0;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads) reduction(+ : nerr)
#endif
    {
        orc_filter *f = (orc_filter *)calloc(1, sizeof(*f)); /* one object per thread, re-initialised per filter */
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long i = 0; i < N; i++) {
            int rc;
            if (kind == ORC_INFORMATION) {
                double I0[NN], i0[ORC_MAXN];
                rc = info_from_state(n, x + i * n, P + i * n * n, i0, I0);
                if (!rc) rc = ldkf_init(f, kind, n, p, 0, i0, I0, F + i * n * n, NULL,
                                        H + i * p * n, Q + i * n * n, R + i * p * p);
            } else {
                rc = ldkf_init(f, kind, n, p, 0, x + i * n, P + i * n * n, F + i * n * n, NULL,
                               H + i * p * n, Q + i * n * n, R + i * p * p);
            }
            if (rc) { nerr++; continue; }
            int bad = 0;
            for (int k = 0; k < T && !bad; k++)
                if (orc_update(f, y + ((long)(k % ypool) * N + i) * p, NULL, NULL, NULL, NULL) != ORC_OK) bad = 1;
            nerr += bad;
            orc_get(f, ORC_GET_STATE, x + i * n);
            orc_get(f, ORC_GET_COVAR, P + i * n * n);
        }
        free(f);
    }
    return nerr;
}

long orc_vanilla_batch(long N, int T, int n, int p,
                       double *x, double *P, const double *F, const double *H,
                       const double *Q, const double *R, const double *y,
                       int threads) {
    return orc_ldkf_batch(ORC_VANILLA, N, T, n, p, x, P, F, H, Q, R, y, T, threads);
}

/* hybrid.go:209-238 HybridKF.SmoothAll / srif.go:165-192 SRIF.SmoothAll (no SNC):
 * for k = l-1 .. 0:  S = inverse(Phi_{k+1});  x_k = S x_{k+1};  P_k = AsSymDense(S P_{k+1} S^T),
 * starting from the last estimate's State() / Covariance().  Phi[k] is the STM stored in estimate k.
 * x[steps][n], P[steps][n*n]: entry steps-1 is input, the others are written. */
int orc_smooth_all(int n, int steps, const double *Phi, double *x, double *P) {
    for (int k = steps - 2; k >= 0; k--) {
        double S[NN], SP[NN], SPSt[NN];
        if (orc_inverse(n, Phi + (size_t)(k + 1) * n * n, S, NULL) != 0) return ORC_ERR_SINGULAR;
        mm(n, n, n, S, P + (size_t)(k + 1) * n * n, SP);
        mm_nt(n, n, n, SP, S, SPSt);
        mv(n, n, S, x + (size_t)(k + 1) * n, x + (size_t)k * n);
        if (orc_as_sym_dense(n, SPSt, P + (size_t)k * n * n) != ORC_OK) return ORC_ERR_ASYMMETRIC;
    }
    return ORC_OK;
}

/* montecarlo.go:18-59 with gonum stat.Mean / stat.StdDev (two-pass, n-1). */
void orc_mc_mean_stddev(long runs, int n, const double *states, double *mean, double *stddev) {
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (long r = 0; r < runs; r++) s += states[r * n + i];
        double mu = s / (double)runs;
        double ss = 0.0, comp = 0.0;
        for (long r = 0; r < runs; r++) {
            double d = states[r * n + i] - mu;
            ss += d * d;
            comp += d;
        }
        double var = (ss - comp * comp / (double)runs) / (double)(runs - 1);
        mean[i] = mu;
        stddev[i] = sqrt(var);
    }
}
