/*
 * TEST INFRASTRUCTURE (CPU oracle) -- not part of the shipped product path.
 *
 * Plain-C restatement of the control-limited iLQR solve of ZhengYi0310/ilqr:
 *   ilqr/ilqr_core.py   (iLQR._dynamics_rollout :69-83, _forward_pass :85-107,
 *                        _backward_pass :109-177, _compute_Q_terms :179-228,
 *                        _trajectory_cost :230-239, update_u :241-399)
 *   ilqr/ilqr_boxQP.py  (boxQP :9-186)
 *   ilqr/ilqr_cost.py   (QRCost :486-632, PathQRCost :635-770)
 *   ilqr/cartpole.py    (CartPoleDynamics :6-95; Jacobian = what T.grad of
 *                        that expression gives, ilqr/diff_utils.py:28-37)
 *   ilqr/cartpole_gym.py (_state_eq :60-72, the closed-loop plant)
 *
 * The restatement keeps the reference's behaviour-defining quirks: the 1/2 in
 * QRCost.l that its gradients do not have, boxQP's un-factorised first
 * iteration (H_ff = ones), the silent `break` on a non-PD Q_uu, the
 * regularised Q_uu/Q_ux in the value update, no clamp in the forward pass,
 * the serial first-hit line search and the mu/delta schedule.
 *
 * Arithmetic conventions (what "bit-exact" means for anything compared with
 * this file): IEEE fp64; the compiler may not contract (-ffp-contract=off).
 * Everything the reference hands to BLAS/LAPACK (np.dot, cholesky, cho_solve)
 * is a left-to-right chain starting with the first product and continued with
 * explicit fma(a, b, acc) -- the instruction the vendor BLAS kernels use on
 * any current CPU -- with matrix products evaluated in the association the
 * reference writes ((f_x^T V_xx) f_x, (f_u^T reg) f_x, (K^T Q_uu) K ...), and
 * triangular solves multiplying by the reciprocal of the diagonal.  Everything
 * the reference evaluates as Python/Theano scalar expressions (the dynamics,
 * cost totals, the mu/delta schedule, line-search ratios, np.sum of an
 * elementwise product) is plain unfused arithmetic in source order.  numpy's
 * BLAS may order the same sums differently; that is a last-bit effect covered
 * by the tolerance tests against the recorded reference runs.
 *
 * Only tests, the smoke check and the benchmark's CPU-baseline leg call this.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "det_math.h"

#define NMAX 64
#define MMAX 16

enum { MODEL_CARTPOLE = 0, MODEL_LINEAR = 1, MODEL_SIGMOID = 2, MODEL_USER = 3 };

/* A user-supplied model (single control): the same source the device library is built from,
 * compiled for the CPU (oracle/user_model_shim.c). */
typedef void (*user_f_fn)(const double *p, const double *x, const double *u, double *out);
typedef void (*user_jac_fn)(const double *p, const double *x, const double *u, double *fx, double *fu);
enum { COST_QR = 0, COST_PATHQR = 1 };
enum { MATH_LIBM = 0, MATH_DET = 1 };

/* Everything that defines one family of problems (shared by a batch). */
typedef struct {
    int model, n, m, T;
    int cost_kind;
    int math_mode;
    int control_limit;
    double dt, mc, mp, l, g;   /* cartpole (cartpole.py:8) */
    const double *A, *B;       /* linear model x+ = A x + B u : [n*n], [n*m] */
    /* synthetic nonlinear model (no counterpart in the reference; it exists to exercise the
     * large-state path): x+ = x + dt (A x + B u + c phi(W x)), phi(z) = z / sqrt(1 + z^2),
     * applied element-wise.  W [n*n], c = sig_c, dt as above. */
    const double *W;
    double sig_c;
    user_f_fn user_f;       /* MODEL_USER */
    user_jac_fn user_jac;
    const double *user_p;   /* its parameters */
    const double *Q, *R, *Qt;  /* [n*n], [m*m], [n*n] */
    const double *x_goal;      /* QR: [n]; PathQR: [(T+1)*n] */
    const double *u_goal;      /* QR: [m]; PathQR: [T*m] */
    const double *lower, *upper; /* [m] */
} oracle_model;

typedef struct {
    int max_iter;        /* n_interations */
    double tol;          /* tolerance */
    int n_alpha;         /* line_search_steps */
    const double *alphas;/* 1.1 ** (-arange(n_alpha)**2), computed by the caller */
    double scale;
    double mu_min, mu_max, delta0;
    /* boxQP options as passed at ilqr_core.py:154-155 */
    int qp_max_iter;
    double qp_min_grad, qp_min_rel_improve, qp_step_dec, qp_min_step, qp_armijo;
} oracle_opts;

/* ------------------------------------------------------------------ math */

static void trig_sincos(int mode, double x, double *s, double *c)
{
    if (mode == MATH_DET) {
        dm_sincos(x, s, c);
    } else {
        *s = sin(x);
        *c = cos(x);
    }
}

static double trig_atan2(int mode, double y, double x)
{
    return mode == MATH_DET ? dm_atan2(y, x) : atan2(y, x);
}

/* ------------------------------------------------------------ dynamics */

/* CartPoleDynamics expression, cartpole.py:72-92, state [x, xd, s, c, thd]. */
static void cartpole_f(const oracle_model *P, const double *x, const double *u, double *out)
{
    double mc = P->mc, mp = P->mp, l = P->l, g = P->g, dt = P->dt;
    double xd = x[1], s = x[2], c = x[3], thd = x[4], F = u[0];
    double M = mc + mp;
    double temp = (F + mp * l * (thd * thd) * s) / M;
    double num = g * s - c * temp;
    double den = l * (4.0 / 3.0 - mp * (c * c) / M);
    double thdd = num / den;
    double xdd = temp - mp * l * thdd * c / M;
    double th = trig_atan2(P->math_mode, s, c);
    double nth = th + thd * dt;
    double sn, cn;
    trig_sincos(P->math_mode, nth, &sn, &cn);
    out[0] = x[0] + xd * dt;
    out[1] = xd + xdd * dt;
    out[2] = sn;
    out[3] = cn;
    out[4] = thd + thdd * dt;
}

/* Jacobian of the expression above with sin_theta, cos_theta treated as
 * independent inputs (that is what T.grad does, diff_utils.py:28-37):
 * fx[5][5] row-major, fu[5][1]. */
static void cartpole_fx_fu(const oracle_model *P, const double *x, const double *u, double *fx,
                           double *fu)
{
    double mc = P->mc, mp = P->mp, l = P->l, g = P->g, dt = P->dt;
    double s = x[2], c = x[3], thd = x[4], F = u[0];
    double M = mc + mp;
    double mpl = mp * l;
    /* The Jacobian expression is this file's own choice (Theano's optimised
     * graph is not observable): reciprocals of M, den and s^2+c^2 are formed
     * once and multiplied, the way a graph optimiser shares sub-expressions. */
    double invM = 1.0 / M;
    double temp = (F + mpl * (thd * thd) * s) * invM;
    double num = g * s - c * temp;
    double den = l * (4.0 / 3.0 - mp * (c * c) * invM);
    double inv_den = 1.0 / den;
    double thdd = num * inv_den;
    /* partials of temp */
    double temp_s = mpl * (thd * thd) * invM;
    double temp_thd = 2.0 * mpl * thd * s * invM;
    double temp_F = invM;
    /* partials of num, den */
    double num_s = g - c * temp_s;
    double num_c = -temp;
    double num_thd = -(c * temp_thd);
    double num_F = -(c * temp_F);
    double den_c = -(2.0 * l * mp * c * invM);
    /* partials of thdd = num / den */
    double thdd_s = num_s * inv_den;
    double thdd_c = (num_c - thdd * den_c) * inv_den;
    double thdd_thd = num_thd * inv_den;
    double thdd_F = num_F * inv_den;
    /* partials of xdd = temp - mpl * thdd * c / M */
    double mplM = mpl * invM;
    double xdd_s = temp_s - mplM * (thdd_s * c);
    double xdd_c = -(mplM * (thdd_c * c + thdd));
    double xdd_thd = temp_thd - mplM * (thdd_thd * c);
    double xdd_F = temp_F - mplM * (thdd_F * c);
    /* theta = atan2(s, c), next_theta = theta + thd*dt */
    double r2 = s * s + c * c;
    double inv_r2 = 1.0 / r2;
    double th_s = c * inv_r2;
    double th_c = -(s * inv_r2);
    double th = trig_atan2(P->math_mode, s, c);
    double nth = th + thd * dt;
    double sn, cn;
    trig_sincos(P->math_mode, nth, &sn, &cn);

    for (int i = 0; i < 25; ++i)
        fx[i] = 0.0;
    fx[0 * 5 + 0] = 1.0;
    fx[0 * 5 + 1] = dt;
    fx[1 * 5 + 1] = 1.0;
    fx[1 * 5 + 2] = xdd_s * dt;
    fx[1 * 5 + 3] = xdd_c * dt;
    fx[1 * 5 + 4] = xdd_thd * dt;
    fx[2 * 5 + 2] = cn * th_s;
    fx[2 * 5 + 3] = cn * th_c;
    fx[2 * 5 + 4] = cn * dt;
    fx[3 * 5 + 2] = -(sn * th_s);
    fx[3 * 5 + 3] = -(sn * th_c);
    fx[3 * 5 + 4] = -(sn * dt);
    fx[4 * 5 + 2] = thdd_s * dt;
    fx[4 * 5 + 3] = thdd_c * dt;
    fx[4 * 5 + 4] = 1.0 + thdd_thd * dt;
    fu[0] = 0.0;
    fu[1] = xdd_F * dt;
    fu[2] = 0.0;
    fu[3] = 0.0;
    fu[4] = thdd_F * dt;
}

static void model_f(const oracle_model *P, const double *x, const double *u, double *out)
{
    if (P->model == MODEL_CARTPOLE) {
        cartpole_f(P, x, u, out);
        return;
    }
    int n = P->n, m = P->m;
    if (P->model == MODEL_USER) {
        P->user_f(P->user_p, x, u, out);
        return;
    }
    if (P->model == MODEL_SIGMOID) {
        for (int i = 0; i < n; ++i) {
            double ax = P->A[i * n + 0] * x[0], z = P->W[i * n + 0] * x[0];
            for (int j = 1; j < n; ++j) {
                ax = fma(P->A[i * n + j], x[j], ax);
                z = fma(P->W[i * n + j], x[j], z);
            }
            double bu = P->B[i * m + 0] * u[0];
            for (int j = 1; j < m; ++j)
                bu = fma(P->B[i * m + j], u[j], bu);
            double s = 1.0 + z * z;
            double phi = z / sqrt(s);
            out[i] = x[i] + P->dt * ((ax + bu) + P->sig_c * phi);
        }
        return;
    }
    for (int i = 0; i < n; ++i) {
        double acc = P->A[i * n + 0] * x[0];
        for (int j = 1; j < n; ++j)
            acc = fma(P->A[i * n + j], x[j], acc);
        for (int j = 0; j < m; ++j)
            acc = fma(P->B[i * m + j], u[j], acc);
        out[i] = acc;
    }
}

static void model_fx_fu(const oracle_model *P, const double *x, const double *u, double *fx,
                        double *fu)
{
    if (P->model == MODEL_CARTPOLE) {
        cartpole_fx_fu(P, x, u, fx, fu);
        return;
    }
    if (P->model == MODEL_USER) {
        P->user_jac(P->user_p, x, u, fx, fu);
        return;
    }
    if (P->model == MODEL_SIGMOID) {
        /* f_x = I + dt (A + c diag(phi'(W x)) W), phi'(z) = 1 / ((1 + z^2) sqrt(1 + z^2)); f_u = dt B */
        int n = P->n, m = P->m;
        for (int i = 0; i < n; ++i) {
            double z = P->W[i * n + 0] * x[0];
            for (int j = 1; j < n; ++j)
                z = fma(P->W[i * n + j], x[j], z);
            double s = 1.0 + z * z;
            double gi = P->sig_c * (1.0 / (s * sqrt(s)));
            for (int j = 0; j < n; ++j)
                fx[i * n + j] = (i == j ? 1.0 : 0.0) + P->dt * (P->A[i * n + j] + gi * P->W[i * n + j]);
            for (int j = 0; j < m; ++j)
                fu[i * m + j] = P->dt * P->B[i * m + j];
        }
        return;
    }
    memcpy(fx, P->A, sizeof(double) * P->n * P->n);
    memcpy(fu, P->B, sizeof(double) * P->n * P->m);
}

/* ---------------------------------------------------------------- cost */

static const double *goal_x(const oracle_model *P, int i)
{
    return P->cost_kind == COST_PATHQR ? P->x_goal + (size_t)i * P->n : P->x_goal;
}
static const double *goal_u(const oracle_model *P, int i)
{
    return P->cost_kind == COST_PATHQR ? P->u_goal + (size_t)i * P->m : P->u_goal;
}

/* v^T M v evaluated as (v^T M) . v, the order of x_diff.T.dot(Q).dot(x_diff). */
static double quad_form(int n, const double *v, const double *M)
{
    double tot = 0.0;
    for (int j = 0; j < n; ++j) {
        double acc = v[0] * M[0 * n + j];
        for (int i = 1; i < n; ++i)
            acc = fma(v[i], M[i * n + j], acc);
        if (j == 0)
            tot = acc * v[0];
        else
            tot = fma(acc, v[j], tot);
    }
    return tot;
}

/* QRCost.l (ilqr_cost.py:532-553, value carries 1/2) and PathQRCost.l
 * (:682-701, no 1/2). */
static double cost_l(const oracle_model *P, const double *x, const double *u, int i, int terminal)
{
    int n = P->n, m = P->m;
    double xd[NMAX], ud[MMAX];
    const double *xg = goal_x(P, i);
    for (int a = 0; a < n; ++a)
        xd[a] = x[a] - xg[a];
    double half = P->cost_kind == COST_QR ? 0.5 : 1.0;
    double sx = quad_form(n, xd, terminal ? P->Qt : P->Q);
    if (P->cost_kind == COST_QR)
        sx = sx * half;
    if (terminal)
        return sx;
    const double *ug = goal_u(P, i);
    for (int a = 0; a < m; ++a)
        ud[a] = u[a] - ug[a];
    double su = quad_form(m, ud, P->R);
    if (P->cost_kind == COST_QR)
        su = su * half;
    return sx + su;
}

/* l_x = x_diff^T (Q + Q^T)   (ilqr_cost.py:555-569) */
static void cost_lx(const oracle_model *P, const double *x, int i, int terminal, double *out)
{
    int n = P->n;
    const double *Qm = terminal ? P->Qt : P->Q;
    const double *xg = goal_x(P, i);
    double xd[NMAX];
    for (int a = 0; a < n; ++a)
        xd[a] = x[a] - xg[a];
    for (int j = 0; j < n; ++j) {
        double acc = xd[0] * (Qm[0 * n + j] + Qm[j * n + 0]);
        for (int a = 1; a < n; ++a)
            acc = fma(xd[a], Qm[a * n + j] + Qm[j * n + a], acc);
        out[j] = acc;
    }
}

/* l_u = u_diff^T (R + R^T)   (ilqr_cost.py:571-587) */
static void cost_lu(const oracle_model *P, const double *u, int i, double *out)
{
    int m = P->m;
    const double *ug = goal_u(P, i);
    double ud[MMAX];
    for (int a = 0; a < m; ++a)
        ud[a] = u[a] - ug[a];
    for (int j = 0; j < m; ++j) {
        double acc = ud[0] * (P->R[0 * m + j] + P->R[j * m + 0]);
        for (int a = 1; a < m; ++a)
            acc = fma(ud[a], P->R[a * m + j] + P->R[j * m + a], acc);
        out[j] = acc;
    }
}

static void cost_lxx(const oracle_model *P, int terminal, double *out)
{
    int n = P->n;
    const double *Qm = terminal ? P->Qt : P->Q;
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
            out[a * n + b] = Qm[a * n + b] + Qm[b * n + a];
}

static void cost_luu(const oracle_model *P, double *out)
{
    int m = P->m;
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b)
            out[a * m + b] = P->R[a * m + b] + P->R[b * m + a];
}

/* ------------------------------------------------------ rollout and J */

/* _dynamics_rollout, ilqr_core.py:69-83 */
static void rollout(const oracle_model *P, const double *x0, const double *us, double *xs)
{
    int n = P->n, m = P->m;
    memcpy(xs, x0, sizeof(double) * n);
    for (int i = 0; i < P->T; ++i)
        model_f(P, xs + (size_t)i * n, us + (size_t)i * m, xs + (size_t)(i + 1) * n);
}

/* _trajectory_cost, ilqr_core.py:230-239: sum() from t = 0 then the terminal term */
static double trajectory_cost(const oracle_model *P, const double *xs, const double *us)
{
    int n = P->n, m = P->m, T = P->T;
    double J = 0.0;
    for (int i = 0; i < T; ++i)
        J = J + cost_l(P, xs + (size_t)i * n, us + (size_t)i * m, i, 0);
    return J + cost_l(P, xs + (size_t)T * n, NULL, T, 1);
}

/* _forward_pass, ilqr_core.py:85-107 (no clamp on us_new) */
static void forward_pass(const oracle_model *P, const double *xs, const double *us, const double *k,
                         const double *K, double alpha, double *xs_new, double *us_new)
{
    int n = P->n, m = P->m;
    memcpy(xs_new, xs, sizeof(double) * n);
    for (int i = 0; i < P->T; ++i) {
        const double *xn = xs_new + (size_t)i * n;
        const double *xo = xs + (size_t)i * n;
        double dx[NMAX];
        for (int a = 0; a < n; ++a)
            dx[a] = xn[a] - xo[a];
        for (int j = 0; j < m; ++j) {
            const double *Kr = K + ((size_t)i * m + j) * n;
            double acc = Kr[0] * dx[0];
            for (int a = 1; a < n; ++a)
                acc = fma(Kr[a], dx[a], acc);
            us_new[(size_t)i * m + j] = (us[(size_t)i * m + j] + alpha * k[(size_t)i * m + j]) + acc;
        }
        model_f(P, xn, us_new + (size_t)i * m, xs_new + (size_t)(i + 1) * n);
    }
}

/* -------------------------------------------------------------- linalg */

/* Lower Cholesky of the leading d x d block stored with row stride ld.
 * Reads the lower triangle only.  Returns 0 on success, 1 on a non-positive
 * (or NaN) pivot. */
static int chol_lower(int d, const double *A, int lda, double *L, int ldl)
{
    for (int j = 0; j < d; ++j) {
        double s = A[j * lda + j];
        for (int k = 0; k < j; ++k)
            s = fma(-L[j * ldl + k], L[j * ldl + k], s);
        if (!(s > 0.0))
            return 1;
        double d_jj = sqrt(s);
        double r_jj = 1.0 / d_jj; /* LAPACK dpotf2 scales the column by ONE / AJJ */
        L[j * ldl + j] = d_jj;
        for (int i = j + 1; i < d; ++i) {
            double t = A[i * lda + j];
            for (int k = 0; k < j; ++k)
                t = fma(-L[i * ldl + k], L[j * ldl + k], t);
            L[i * ldl + j] = t * r_jj;
        }
    }
    return 0;
}

/* Solve (L L^T) x = b in place; L lower, row stride ldl (cho_solve((L, True), b)).
 * Each substitution multiplies by the reciprocal of the diagonal entry, which is
 * how the triangular-solve kernels of the BLAS behind scipy (OpenBLAS packs the
 * triangle with inverted diagonal) do it. */
static void chol_solve(int d, const double *L, int ldl, double *b)
{
    for (int i = 0; i < d; ++i) {
        double s = b[i];
        for (int j = 0; j < i; ++j)
            s = fma(-L[i * ldl + j], b[j], s);
        b[i] = s * (1.0 / L[i * ldl + i]);
    }
    for (int i = d - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < d; ++j)
            s = fma(-L[j * ldl + i], b[j], s);
        b[i] = s * (1.0 / L[i * ldl + i]);
    }
}

static int all_finite(int cnt, const double *a)
{
    for (int i = 0; i < cnt; ++i)
        if (!isfinite(a[i]))
            return 0;
    return 1;
}

/* --------------------------------------------------------------- boxQP */

typedef struct {
    int n_iter;      /* main-loop passes that reached the gradient step */
    int n_factor;    /* factorisations */
    int n_backtrack; /* Armijo step reductions, summed */
} qp_counters;

static double qp_value(int m, const double *H, const double *g, const double *x)
{
    /* np.dot(x.T, g) + 0.5 * np.dot(np.dot(x.T, H), x)   ilqr_boxQP.py:62 */
    double xg = x[0] * g[0];
    for (int i = 1; i < m; ++i)
        xg = fma(x[i], g[i], xg);
    return xg + 0.5 * quad_form(m, x, H);
}

static double clampd(double lo, double hi, double v)
{
    /* np.maximum(lower, np.minimum(upper, x)) with NaN propagation */
    double t = (v != v) ? v : (hi < v ? hi : v);
    return (t != t) ? t : (lo > t ? lo : t);
}

/* boxQP, ilqr_boxQP.py:9-186.  Hff receives the free-subspace Cholesky factor
 * packed as a (hff_dim x hff_dim) row-major matrix with row stride m; when no
 * factorisation ever happened it is the all-ones m x m matrix of :36.
 * Returns 0, or -1 when a LinAlgError would have been raised (non-finite H). */
static int boxqp(int m, const double *H, const double *g, const double *lower, const double *upper,
                 const double *x0, const oracle_opts *O, double *x, int *result_out, double *Hff,
                 int *hff_dim, int *free_out, qp_counters *cnt)
{
    int clamped[MMAX], freev[MMAX], oldclamped[MMAX];
    double grad[MMAX], search[MMAX], xc[MMAX], rhs[MMAX];
    double oldvalue = 0.0, value;
    int result = -1;
    int hd = m;
    for (int i = 0; i < m; ++i) {
        clamped[i] = 0;
        freev[i] = 0;
    }
    for (int i = 0; i < m * m; ++i)
        Hff[i] = 1.0;
    if (x0) {
        for (int i = 0; i < m; ++i)
            x[i] = clampd(lower[i], upper[i], x0[i]);
    } else {
        for (int i = 0; i < m; ++i)
            x[i] = (lower[i] + upper[i]) / 2;
    }
    value = qp_value(m, H, g, x);

    /* np.all(np.linalg.eigvals(H) > 0): H never changes inside the loop */
    int h_finite = all_finite(m * m, H);
    double Lfull[MMAX * MMAX];
    int h_pd = h_finite && chol_lower(m, H, m, Lfull, m) == 0;

    for (int iter = 0; iter < O->qp_max_iter; ++iter) {
        if (result != -1)
            break;
        if (iter > 1 && (oldvalue - value) < O->qp_min_rel_improve * fabs(oldvalue) &&
            (oldvalue - value) > 0) {
            result = 4;
            break;
        }
        oldvalue = value;
        for (int i = 0; i < m; ++i) {
            double acc = H[i * m + 0] * x[0];
            for (int j = 1; j < m; ++j)
                acc = fma(H[i * m + j], x[j], acc);
            grad[i] = g[i] + acc;
        }
        int nclamped = 0;
        for (int i = 0; i < m; ++i) {
            oldclamped[i] = clamped[i];
            clamped[i] = ((x[i] == lower[i]) && (grad[i] > 0)) || ((x[i] == upper[i]) && (grad[i] < 0));
            freev[i] = !clamped[i];
            nclamped += clamped[i];
        }
        if (nclamped == m) {
            result = 6;
            break;
        }
        if (!h_finite)
            return -1; /* eigvals raises LinAlgError */
        if (!h_pd) {
            result = -1;
            break;
        }
        int factorize = 0;
        if (iter == 1) {
            factorize = 1;
        } else {
            for (int i = 0; i < m; ++i)
                if (oldclamped[i] != clamped[i])
                    factorize = 1;
        }
        int fidx[MMAX], cidx[MMAX], nf = 0, nc = 0;
        for (int i = 0; i < m; ++i) {
            if (freev[i])
                fidx[nf++] = i;
            else
                cidx[nc++] = i;
        }
        if (factorize) {
            double sub[MMAX * MMAX];
            for (int a = 0; a < nf; ++a)
                for (int b = 0; b < nf; ++b)
                    sub[a * m + b] = H[fidx[a] * m + fidx[b]];
            if (chol_lower(nf, sub, m, Hff, m) != 0)
                return -1; /* sla.cholesky raises LinAlgError */
            hd = nf;
            cnt->n_factor++;
        }
        /* sla.norm(grad[free]) */
        double ss = 0.0;
        for (int a = 0; a < nf; ++a)
            ss = (a == 0) ? grad[fidx[0]] * grad[fidx[0]] : fma(grad[fidx[a]], grad[fidx[a]], ss);
        /* dnrm2 of a single element is its magnitude, exactly */
        double gnorm = nf == 1 ? fabs(grad[fidx[0]]) : sqrt(ss);
        if (gnorm < O->qp_min_grad) {
            result = 5;
            break;
        }
        cnt->n_iter++;
        /* grad_clamped = g[free] + H_fc x[clamped] */
        for (int a = 0; a < nf; ++a) {
            double acc = 0.0;
            for (int b = 0; b < nc; ++b)
                acc = (b == 0) ? H[fidx[a] * m + cidx[0]] * x[cidx[0]]
                               : fma(H[fidx[a] * m + cidx[b]], x[cidx[b]], acc);
            rhs[a] = g[fidx[a]] + acc;
        }
        /* hd == nf always holds here: either a factorisation with this free set
         * happened, or nothing is clamped and Hff is still ones(m, m). */
        chol_solve(nf, Hff, m, rhs);
        for (int i = 0; i < m; ++i)
            search[i] = 0.0;
        for (int a = 0; a < nf; ++a)
            search[fidx[a]] = rhs[a] * -1 - x[fidx[a]];

        double step = 1.0;
        double sdotg, vc;
        for (int i = 0; i < m; ++i)
            xc[i] = clampd(lower[i], upper[i], x[i] + step * search[i]);
        sdotg = (xc[0] - x[0]) * grad[0];
        for (int i = 1; i < m; ++i)
            sdotg = sdotg + (xc[i] - x[i]) * grad[i];
        vc = qp_value(m, H, g, xc);
        while ((vc - oldvalue) / (sdotg + 1e-10) < O->qp_armijo) {
            step = step * O->qp_step_dec;
            cnt->n_backtrack++;
            for (int i = 0; i < m; ++i)
                xc[i] = clampd(lower[i], upper[i], x[i] + step * search[i]);
            vc = qp_value(m, H, g, xc);
            sdotg = (xc[0] - x[0]) * grad[0];
            for (int i = 1; i < m; ++i)
                sdotg = sdotg + (xc[i] - x[i]) * grad[i];
            if (step < O->qp_min_step) {
                result = 2;
                break;
            }
        }
        for (int i = 0; i < m; ++i)
            x[i] = xc[i];
        value = vc;
    }
    *result_out = result;
    *hff_dim = hd;
    for (int i = 0; i < m; ++i)
        free_out[i] = freev[i];
    return 0;
}

/* Optional capture of every boxQP call made by the backward pass: rows of
 * (H, g, lower, upper, x0) for m = 1.  Single-threaded diagnostic use only. */
static double *g_qp_capture = NULL;
static long g_qp_capture_cap = 0, g_qp_capture_n = 0;
void oracle_set_qp_capture(double *buf, long cap) { g_qp_capture = buf; g_qp_capture_cap = cap; g_qp_capture_n = 0; }
long oracle_get_qp_capture_count(void) { return g_qp_capture_n; }

/* Test hook: per-iteration history of the per-step boxQP outcomes of ONE problem solved with a trace
 * (single-threaded use): after every iteration's backward pass the result codes [T] and free masks [T*m] are
 * copied to row `iteration` of these buffers (what boxQP returns at ilqr_core.py:156, step by step). */
static int *g_hist_qp = NULL, *g_hist_free = NULL;
static long g_hist_cap = 0;
void oracle_set_step_history(int *qp_result_hist, int *free_hist, long cap_iters)
{
    g_hist_qp = qp_result_hist;
    g_hist_free = free_hist;
    g_hist_cap = cap_iters;
}

/* ------------------------------------------------------- backward pass */

typedef struct {
    double dJ1, dJ2;
    int stopped_at;     /* step index of the non-PD break, -1 if none (ilqr_core.py:137-140) */
    int qp_iter, qp_factor, qp_backtrack; /* summed over steps */
    int n_clamped_steps;                  /* steps with at least one clamped control */
} bw_info;

/* _backward_pass + _compute_Q_terms, ilqr_core.py:109-228.
 * k [T*m], K [T*m*n] are fully overwritten (zeros where the sweep did not reach).
 * Optional per-step outputs: qp_result [T], free_mask [T*m].
 * Returns 0, or -1 where the reference would raise LinAlgError. */
static int backward_pass(const oracle_model *P, const oracle_opts *O, const double *xs,
                         const double *us, double mu, double *k, double *K, bw_info *info,
                         int *qp_result, int *free_mask)
{
    int n = P->n, m = P->m, T = P->T;
    double Vx[NMAX], Vxx[NMAX * NMAX];
    double fx[NMAX * NMAX], fu[NMAX * MMAX];
    double lx[NMAX], lu[MMAX], lxx[NMAX * NMAX], luu[MMAX * MMAX];
    double Qx[NMAX], Qu[MMAX], Qxx[NMAX * NMAX], Qux[MMAX * NMAX], Quu[MMAX * MMAX];
    double M1[NMAX * NMAX], reg[NMAX * NMAX], M2[MMAX * NMAX];
    double L[MMAX * MMAX], Hff[MMAX * MMAX];
    double T1[NMAX * MMAX];

    info->dJ1 = 0.0;
    info->dJ2 = 0.0;
    info->stopped_at = -1;
    info->qp_iter = info->qp_factor = info->qp_backtrack = 0;
    info->n_clamped_steps = 0;
    memset(k, 0, sizeof(double) * T * m);
    memset(K, 0, sizeof(double) * T * m * n);
    if (qp_result)
        for (int i = 0; i < T; ++i)
            qp_result[i] = -100; /* step not reached */
    if (free_mask)
        memset(free_mask, 0, sizeof(int) * T * m);

    cost_lx(P, xs + (size_t)T * n, T, 1, Vx);
    cost_lxx(P, 1, Vxx);

    for (int i = T - 1; i >= 0; --i) {
        const double *x = xs + (size_t)i * n;
        const double *u = us + (size_t)i * m;
        double *ki = k + (size_t)i * m;
        double *Ki = K + (size_t)i * m * n;

        model_fx_fu(P, x, u, fx, fu);
        cost_lx(P, x, i, 0, lx);
        cost_lu(P, u, i, lu);
        cost_lxx(P, 0, lxx);
        cost_luu(P, luu);
        /* l_ux is identically zero (ilqr_cost.py:603-615) */

        for (int j = 0; j < n; ++j) {
            double acc = fx[0 * n + j] * Vx[0];
            for (int a = 1; a < n; ++a)
                acc = fma(fx[a * n + j], Vx[a], acc);
            Qx[j] = lx[j] + acc;
        }
        for (int j = 0; j < m; ++j) {
            double acc = fu[0 * m + j] * Vx[0];
            for (int a = 1; a < n; ++a)
                acc = fma(fu[a * m + j], Vx[a], acc);
            Qu[j] = lu[j] + acc;
        }
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                double acc = fx[0 * n + a] * Vxx[0 * n + b];
                for (int c = 1; c < n; ++c)
                    acc = fma(fx[c * n + a], Vxx[c * n + b], acc);
                M1[a * n + b] = acc;
            }
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                double acc = M1[a * n + 0] * fx[0 * n + b];
                for (int c = 1; c < n; ++c)
                    acc = fma(M1[a * n + c], fx[c * n + b], acc);
                Qxx[a * n + b] = lxx[a * n + b] + acc;
            }
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b)
                reg[a * n + b] = Vxx[a * n + b] + mu * (a == b ? 1.0 : 0.0);
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < n; ++b) {
                double acc = fu[0 * m + a] * reg[0 * n + b];
                for (int c = 1; c < n; ++c)
                    acc = fma(fu[c * m + a], reg[c * n + b], acc);
                M2[a * n + b] = acc;
            }
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < n; ++b) {
                double acc = M2[a * n + 0] * fx[0 * n + b];
                for (int c = 1; c < n; ++c)
                    acc = fma(M2[a * n + c], fx[c * n + b], acc);
                Qux[a * n + b] = 0.0 + acc;
            }
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) {
                double acc = M2[a * n + 0] * fu[0 * m + b];
                for (int c = 1; c < n; ++c)
                    acc = fma(M2[a * n + c], fu[c * m + b], acc);
                Quu[a * m + b] = luu[a * m + b] + acc;
            }

        /* np.all(np.linalg.eigvals(Q_uu) > 0) else print/warn/break */
        if (!all_finite(m * m, Quu))
            return -1; /* eigvals raises LinAlgError on NaN/Inf */
        if (chol_lower(m, Quu, m, L, m) != 0) {
            info->stopped_at = i;
            break;
        }
        /* k[i] = cho_solve((R_Q_uu, True), Q_u)  -- boxQP warm start (positive sign) */
        for (int a = 0; a < m; ++a)
            ki[a] = Qu[a];
        chol_solve(m, L, m, ki);

        if (P->control_limit) {
            double lo[MMAX], hi[MMAX], xq[MMAX];
            int result, hd, freev[MMAX];
            qp_counters cnt = {0, 0, 0};
            for (int a = 0; a < m; ++a) {
                lo[a] = P->lower[a] - u[a];
                hi[a] = P->upper[a] - u[a];
            }
            if (g_qp_capture && m == 1 && g_qp_capture_n < g_qp_capture_cap) {
                double *row = g_qp_capture + 5 * g_qp_capture_n++;
                row[0] = Quu[0]; row[1] = Qu[0]; row[2] = lo[0]; row[3] = hi[0]; row[4] = ki[0];
            }
            if (boxqp(m, Quu, Qu, lo, hi, ki, O, xq, &result, Hff, &hd, freev, &cnt) != 0)
                return -1;
            for (int a = 0; a < m; ++a)
                ki[a] = xq[a];
            info->qp_iter += cnt.n_iter;
            info->qp_factor += cnt.n_factor;
            info->qp_backtrack += cnt.n_backtrack;
            int nf = 0, fidx[MMAX];
            for (int a = 0; a < m; ++a)
                if (freev[a])
                    fidx[nf++] = a;
            if (nf < m)
                info->n_clamped_steps++;
            if (qp_result)
                qp_result[i] = result;
            if (free_mask)
                for (int a = 0; a < m; ++a)
                    free_mask[(size_t)i * m + a] = freev[a];
            if (nf > 0) {
                /* K[i, free, :] = -cho_solve((H_ff, True), Q_ux[free, :]) */
                double col[MMAX];
                for (int b = 0; b < n; ++b) {
                    for (int a = 0; a < nf; ++a)
                        col[a] = Qux[fidx[a] * n + b];
                    chol_solve(nf, Hff, m, col);
                    for (int a = 0; a < nf; ++a)
                        Ki[fidx[a] * n + b] = col[a] * -1;
                }
            }
        } else {
            /* K[i] = cho_solve((R_Q_uu, True), Q_ux): the wrong-sign branch, kept as is */
            double col[MMAX];
            for (int b = 0; b < n; ++b) {
                for (int a = 0; a < m; ++a)
                    col[a] = Qux[a * n + b];
                chol_solve(m, L, m, col);
                for (int a = 0; a < m; ++a)
                    Ki[a * n + b] = col[a];
            }
        }

        /* T1 = K^T Q_uu  [n x m] */
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < m; ++b) {
                double acc = Ki[0 * n + a] * Quu[0 * m + b];
                for (int c = 1; c < m; ++c)
                    acc = fma(Ki[c * n + a], Quu[c * m + b], acc);
                T1[a * m + b] = acc;
            }
        /* V_x = Q_x + (K^T Q_uu) k + K^T Q_u + Q_ux^T k */
        double Vx_new[NMAX];
        for (int a = 0; a < n; ++a) {
            double t1 = T1[a * m + 0] * ki[0];
            double t2 = Ki[0 * n + a] * Qu[0];
            double t3 = Qux[0 * n + a] * ki[0];
            for (int c = 1; c < m; ++c) {
                t1 = fma(T1[a * m + c], ki[c], t1);
                t2 = fma(Ki[c * n + a], Qu[c], t2);
                t3 = fma(Qux[c * n + a], ki[c], t3);
            }
            Vx_new[a] = ((Qx[a] + t1) + t2) + t3;
        }
        /* V_xx = Q_xx + (K^T Q_uu) K + K^T Q_ux + Q_ux^T K ; then symmetrise */
        double Vn[NMAX * NMAX];
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                double t1 = T1[a * m + 0] * Ki[0 * n + b];
                double t2 = Ki[0 * n + a] * Qux[0 * n + b];
                double t3 = Qux[0 * n + a] * Ki[0 * n + b];
                for (int c = 1; c < m; ++c) {
                    t1 = fma(T1[a * m + c], Ki[c * n + b], t1);
                    t2 = fma(Ki[c * n + a], Qux[c * n + b], t2);
                    t3 = fma(Qux[c * n + a], Ki[c * n + b], t3);
                }
                Vn[a * n + b] = ((Qxx[a * n + b] + t1) + t2) + t3;
            }
        for (int a = 0; a < n; ++a)
            Vx[a] = Vx_new[a];
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b)
                Vxx[a * n + b] = 0.5 * (Vn[a * n + b] + Vn[b * n + a]);

        /* Delta_J1 += k.Q_u ; Delta_J2 += 0.5 * k.(Q_uu k) */
        double d1 = ki[0] * Qu[0];
        for (int c = 1; c < m; ++c)
            d1 = fma(ki[c], Qu[c], d1);
        double d2 = 0.0;
        for (int a = 0; a < m; ++a) {
            double acc = Quu[a * m + 0] * ki[0];
            for (int c = 1; c < m; ++c)
                acc = fma(Quu[a * m + c], ki[c], acc);
            d2 = (a == 0) ? ki[0] * acc : fma(ki[a], acc, d2);
        }
        info->dJ1 = info->dJ1 + d1;
        info->dJ2 = info->dJ2 + d2 * 0.5;
    }
    return 0;
}

/* ------------------------------------------------------------ update_u */

/* Per-iteration trace (arrays of length max_iter, any pointer may be NULL). */
typedef struct {
    double *J;          /* J_opt after the iteration */
    int *accepted;
    int *alpha_idx;     /* index at which the line search ended; n_alpha if it ran out */
    double *mu, *delta; /* after the schedule update */
    double *ratio;      /* (J - J_new) / expected at the last evaluated alpha */
    int *qp_iter, *qp_backtrack, *n_clamped, *stopped_at;
} oracle_trace;

typedef struct {
    double J;
    double mu, delta;
    int iters;      /* iterations started */
    int converged;
    int status;     /* 0 ran out of iterations, 1 converged, 2 mu >= mu_max */
} oracle_result;

/* iLQR.update_u, ilqr_core.py:241-399.
 * delta_io: the solver's delta_ that persists across calls (:62, :317, :335).
 * k_io [T*m], K_io [T*m*n]: the solver's k_/K_ (zeros at construction :64-65).
 * xs_out [(T+1)*n], us_out [T*m]. */
static void update_u(const oracle_model *P, const oracle_opts *O, const double *x0,
                     const double *us_init, double *delta_io, double *k_io, double *K_io,
                     double *xs_out, double *us_out, oracle_result *res, const oracle_trace *tr,
                     int *last_qp_result, int *last_free)
{
    int n = P->n, m = P->m, T = P->T;
    size_t nx = (size_t)(T + 1) * n, nu = (size_t)T * m, nK = (size_t)T * m * n;
    double *xs = (double *)malloc(sizeof(double) * nx);
    double *us = (double *)malloc(sizeof(double) * nu);
    double *xs_new = (double *)calloc(nx, sizeof(double));
    double *us_new = (double *)calloc(nu, sizeof(double));
    double *kt = (double *)malloc(sizeof(double) * nu);
    double *Kt = (double *)malloc(sizeof(double) * nK);
    double *k = k_io, *K = K_io;

    double mu = 1.0; /* :264 */
    double delta = *delta_io;
    memcpy(us, us_init, sizeof(double) * nu);
    rollout(P, x0, us, xs);
    double J = trajectory_cost(P, xs, us);
    double dJ1 = 0.0, dJ2 = 0.0, actual = 0.0;
    int converged = 0, status = 0, it = 0;

    for (it = 0; it < O->max_iter; ++it) {
        int accepted = 0;
        bw_info info;
        info.stopped_at = -1;
        info.qp_iter = info.qp_backtrack = info.n_clamped_steps = 0;
        int done = 0;
        while (!done) {
            int rc = backward_pass(P, O, xs, us, mu, kt, Kt, &info, last_qp_result, last_free);
            if (rc == 0) {
                memcpy(k, kt, sizeof(double) * nu);
                memcpy(K, Kt, sizeof(double) * nK);
                dJ1 = info.dJ1;
                dJ2 = info.dJ2;
                done = 1;
            } else {
                /* except LinAlgError  :291-299 */
                delta = fmax(delta, 1.0) * O->delta0;
                mu = fmax(O->mu_min, mu * delta);
                if (O->mu_max != 0.0 && mu >= O->mu_max)
                    break;
            }
        }
        if (g_hist_qp && g_hist_free && last_qp_result && last_free && it < g_hist_cap) {
            memcpy(g_hist_qp + (size_t)it * T, last_qp_result, sizeof(int) * T);
            memcpy(g_hist_free + (size_t)it * T * m, last_free, sizeof(int) * T * m);
        }
        int j;
        double ratio = NAN, Jnew = 0.0;
        for (j = 0; j < O->n_alpha; ++j) {
            double alpha = O->alphas[j];
            forward_pass(P, xs, us, k, K, alpha, xs_new, us_new);
            Jnew = trajectory_cost(P, xs_new, us_new);
            double expected = (alpha * dJ1 + alpha * alpha * dJ2) * -1;
            if (expected <= 0) {
                break;
            } else {
                ratio = (J - Jnew) / expected;
                if (ratio > O->scale) {
                    actual = J - Jnew;
                    accepted = 1;
                    break;
                }
            }
        }
        int stop = 0;
        if (accepted) {
            delta = fmin(1.0, delta) / O->delta0;
            mu = mu * delta;
            if (mu <= O->mu_min)
                mu = 0;
            J = Jnew;
            memcpy(xs, xs_new, sizeof(double) * nx);
            memcpy(us, us_new, sizeof(double) * nu);
            if (actual < O->tol)
                converged = 1;
        } else {
            delta = fmax(delta, 1.0) * O->delta0;
            mu = fmax(O->mu_min, mu * delta);
            if (O->mu_max != 0.0 && mu >= O->mu_max) {
                status = 2;
                stop = 1;
            }
        }
        if (tr) {
            if (tr->J) tr->J[it] = J;
            if (tr->accepted) tr->accepted[it] = accepted;
            if (tr->alpha_idx) tr->alpha_idx[it] = j;
            if (tr->mu) tr->mu[it] = mu;
            if (tr->delta) tr->delta[it] = delta;
            if (tr->ratio) tr->ratio[it] = ratio;
            if (tr->qp_iter) tr->qp_iter[it] = info.qp_iter;
            if (tr->qp_backtrack) tr->qp_backtrack[it] = info.qp_backtrack;
            if (tr->n_clamped) tr->n_clamped[it] = info.n_clamped_steps;
            if (tr->stopped_at) tr->stopped_at[it] = info.stopped_at;
        }
        if (stop) {
            ++it;
            break;
        }
        if (converged) {
            status = 1;
            ++it;
            break;
        }
    }
    *delta_io = delta;
    memcpy(xs_out, xs, sizeof(double) * nx);
    memcpy(us_out, us, sizeof(double) * nu);
    res->J = J;
    res->mu = mu;
    res->delta = delta;
    res->iters = it;
    res->converged = converged;
    res->status = status;
    free(xs);
    free(us);
    free(xs_new);
    free(us_new);
    free(kt);
    free(Kt);
}

/* ============================================================ exported */

/* Flat description used across the ctypes boundary. */
typedef struct {
    int model, n, m, T, cost_kind, math_mode, control_limit, pad_;
    double dt, mc, mp, l, g;
    const double *A, *B, *Q, *R, *Qt, *x_goal, *u_goal, *lower, *upper;
    const double *W;
    double sig_c;
    user_f_fn user_f;
    user_jac_fn user_jac;
    const double *user_p;
} oracle_model_c;

static oracle_model to_model(const oracle_model_c *c)
{
    oracle_model P;
    P.model = c->model; P.n = c->n; P.m = c->m; P.T = c->T;
    P.cost_kind = c->cost_kind; P.math_mode = c->math_mode; P.control_limit = c->control_limit;
    P.dt = c->dt; P.mc = c->mc; P.mp = c->mp; P.l = c->l; P.g = c->g;
    P.A = c->A; P.B = c->B; P.Q = c->Q; P.R = c->R; P.Qt = c->Qt;
    P.x_goal = c->x_goal; P.u_goal = c->u_goal; P.lower = c->lower; P.upper = c->upper;
    P.W = c->W; P.sig_c = c->sig_c;
    P.user_f = c->user_f; P.user_jac = c->user_jac; P.user_p = c->user_p;
    return P;
}

int oracle_dims_ok(int n, int m) { return n >= 1 && n <= NMAX && m >= 1 && m <= MMAX; }

void oracle_detmath(int cnt, const double *a, const double *b, double *sin_a, double *cos_a,
                    double *atan2_ab)
{
    for (int i = 0; i < cnt; ++i) {
        dm_sincos(a[i], &sin_a[i], &cos_a[i]);
        atan2_ab[i] = dm_atan2(a[i], b[i]);
    }
}

/* f, f_x, f_u at cnt points: x [cnt*n], u [cnt*m] -> f [cnt*n], fx [cnt*n*n], fu [cnt*n*m] */
void oracle_dynamics(const oracle_model_c *c, int cnt, const double *x, const double *u, double *f,
                     double *fx, double *fu)
{
    oracle_model P = to_model(c);
    for (int i = 0; i < cnt; ++i) {
        model_f(&P, x + (size_t)i * P.n, u + (size_t)i * P.m, f + (size_t)i * P.n);
        model_fx_fu(&P, x + (size_t)i * P.n, u + (size_t)i * P.m, fx + (size_t)i * P.n * P.n,
                    fu + (size_t)i * P.n * P.m);
    }
}

/* Per-timestep derivative tensors of one trajectory, in the order the
 * backward pass consumes them: fx [T*n*n], fu [T*n*m], lx [T*n], lu [T*m],
 * lxx [T*n*n], luu [T*m*m], lux [T*m*n], and terminal lx_T [n], lxx_T [n*n]. */
void oracle_derivatives(const oracle_model_c *c, const double *xs, const double *us, double *fx,
                        double *fu, double *lx, double *lu, double *lxx, double *luu, double *lux,
                        double *lxT, double *lxxT)
{
    oracle_model P = to_model(c);
    int n = P.n, m = P.m, T = P.T;
    for (int i = 0; i < T; ++i) {
        model_fx_fu(&P, xs + (size_t)i * n, us + (size_t)i * m, fx + (size_t)i * n * n,
                    fu + (size_t)i * n * m);
        cost_lx(&P, xs + (size_t)i * n, i, 0, lx + (size_t)i * n);
        cost_lu(&P, us + (size_t)i * m, i, lu + (size_t)i * m);
        cost_lxx(&P, 0, lxx + (size_t)i * n * n);
        cost_luu(&P, luu + (size_t)i * m * m);
        for (int a = 0; a < m * n; ++a)
            lux[(size_t)i * m * n + a] = 0.0;
    }
    cost_lx(&P, xs + (size_t)T * n, T, 1, lxT);
    cost_lxx(&P, 1, lxxT);
}

void oracle_rollout_cost(const oracle_model_c *c, int B, const double *x0, const double *us,
                         double *xs, double *J)
{
    oracle_model P = to_model(c);
    size_t nx = (size_t)(P.T + 1) * P.n, nu = (size_t)P.T * P.m;
    for (int b = 0; b < B; ++b) {
        rollout(&P, x0 + (size_t)b * P.n, us + b * nu, xs + b * nx);
        J[b] = trajectory_cost(&P, xs + b * nx, us + b * nu);
    }
}

void oracle_trajectory_cost(const oracle_model_c *c, int B, const double *xs, const double *us,
                            double *J)
{
    oracle_model P = to_model(c);
    size_t nx = (size_t)(P.T + 1) * P.n, nu = (size_t)P.T * P.m;
    for (int b = 0; b < B; ++b)
        J[b] = trajectory_cost(&P, xs + b * nx, us + b * nu);
}

/* One backward pass per problem.  info [B*8]: dJ1, dJ2, stopped_at, qp_iter,
 * qp_factor, qp_backtrack, n_clamped_steps, rc. */
void oracle_backward(const oracle_model_c *c, const oracle_opts *O, int B, const double *xs,
                     const double *us, const double *mu, double *k, double *K, double *info,
                     int *qp_result, int *free_mask)
{
    oracle_model P = to_model(c);
    size_t nx = (size_t)(P.T + 1) * P.n, nu = (size_t)P.T * P.m, nK = nu * P.n;
    for (int b = 0; b < B; ++b) {
        bw_info bi;
        int rc = backward_pass(&P, O, xs + b * nx, us + b * nu, mu[b], k + b * nu, K + b * nK, &bi,
                               qp_result ? qp_result + (size_t)b * P.T : NULL,
                               free_mask ? free_mask + b * nu : NULL);
        double *o = info + (size_t)b * 8;
        o[0] = bi.dJ1; o[1] = bi.dJ2; o[2] = bi.stopped_at; o[3] = bi.qp_iter;
        o[4] = bi.qp_factor; o[5] = bi.qp_backtrack; o[6] = bi.n_clamped_steps; o[7] = rc;
    }
}

/* Forward pass + cost for every (problem, alpha): xs_new [B*A*(T+1)*n], us_new
 * [B*A*T*m], J [B*A]. */
void oracle_forward(const oracle_model_c *c, int B, int nA, const double *alphas, const double *xs,
                    const double *us, const double *k, const double *K, double *xs_new,
                    double *us_new, double *J)
{
    oracle_model P = to_model(c);
    size_t nx = (size_t)(P.T + 1) * P.n, nu = (size_t)P.T * P.m, nK = nu * P.n;
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < nA; ++a) {
            size_t o = (size_t)b * nA + a;
            forward_pass(&P, xs + b * nx, us + b * nu, k + b * nu, K + b * nK, alphas[a],
                         xs_new + o * nx, us_new + o * nu);
            J[o] = trajectory_cost(&P, xs_new + o * nx, us_new + o * nu);
        }
}

/* Standalone boxQP over a batch: H [B*m*m], g/lower/upper/x0 [B*m] (x0 may be
 * NULL) -> x [B*m], result [B], Hff [B*m*m], hff_dim [B], free [B*m],
 * counters [B*3] = (n_iter, n_factor, n_backtrack), rc [B]. */
void oracle_boxqp(int B, int m, const double *H, const double *g, const double *lower,
                  const double *upper, const double *x0, const oracle_opts *O, double *x,
                  int *result, double *Hff, int *hff_dim, int *free_out, int *counters, int *rc)
{
    for (int b = 0; b < B; ++b) {
        qp_counters cnt = {0, 0, 0};
        size_t o = (size_t)b * m;
        rc[b] = boxqp(m, H + o * m, g + o, lower + o, upper + o, x0 ? x0 + o : NULL, O, x + o,
                      &result[b], Hff + o * m, &hff_dim[b], free_out + o, &cnt);
        counters[b * 3 + 0] = cnt.n_iter;
        counters[b * 3 + 1] = cnt.n_factor;
        counters[b * 3 + 2] = cnt.n_backtrack;
    }
}

/* Batched update_u.  Per problem b:
 *   x0 [n], us_init [T*m], delta_io [1], k_io [T*m], K_io [T*m*n]
 *   -> xs [(T+1)*n], us [T*m], res_d [4] = (J, mu, delta, 0), res_i [4] =
 *      (iters, converged, status, 0).
 * Trace arrays are [B*max_iter] (NULL to skip); last_qp_result [B*T],
 * last_free [B*T*m] (NULL to skip).  nthreads > 1 uses OpenMP over problems. */
void oracle_solve(const oracle_model_c *c, const oracle_opts *O, int B, int nthreads,
                  const double *x0, const double *us_init, double *delta_io, double *k_io,
                  double *K_io, double *xs, double *us, double *res_d, int *res_i, double *tr_J,
                  int *tr_accepted, int *tr_alpha, double *tr_mu, double *tr_delta,
                  double *tr_ratio, int *tr_qp_iter, int *tr_qp_backtrack, int *tr_clamped,
                  int *tr_stopped, int *last_qp_result, int *last_free)
{
    oracle_model P = to_model(c);
    size_t nx = (size_t)(P.T + 1) * P.n, nu = (size_t)P.T * P.m, nK = nu * P.n;
    int mi = O->max_iter;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int b = 0; b < B; ++b) {
        oracle_trace tr;
        size_t o = (size_t)b * mi;
        tr.J = tr_J ? tr_J + o : NULL;
        tr.accepted = tr_accepted ? tr_accepted + o : NULL;
        tr.alpha_idx = tr_alpha ? tr_alpha + o : NULL;
        tr.mu = tr_mu ? tr_mu + o : NULL;
        tr.delta = tr_delta ? tr_delta + o : NULL;
        tr.ratio = tr_ratio ? tr_ratio + o : NULL;
        tr.qp_iter = tr_qp_iter ? tr_qp_iter + o : NULL;
        tr.qp_backtrack = tr_qp_backtrack ? tr_qp_backtrack + o : NULL;
        tr.n_clamped = tr_clamped ? tr_clamped + o : NULL;
        tr.stopped_at = tr_stopped ? tr_stopped + o : NULL;
        oracle_result r;
        update_u(&P, O, x0 + (size_t)b * P.n, us_init + b * nu, delta_io + b, k_io + b * nu,
                 K_io + b * nK, xs + b * nx, us + b * nu, &r, &tr,
                 last_qp_result ? last_qp_result + (size_t)b * P.T : NULL,
                 last_free ? last_free + b * nu : NULL);
        res_d[b * 4 + 0] = r.J; res_d[b * 4 + 1] = r.mu; res_d[b * 4 + 2] = r.delta; res_d[b * 4 + 3] = 0;
        res_i[b * 4 + 0] = r.iters; res_i[b * 4 + 1] = r.converged; res_i[b * 4 + 2] = r.status;
        res_i[b * 4 + 3] = 0;
    }
}

/* CartPoleContinuousEnv._state_eq, cartpole_gym.py:60-72 on [x, xd, theta, thd].
 * par = (gravity, masscart, masspole, length, tau). */
void oracle_plant_step(int B, int math_mode, const double *par, const double *st, const double *u,
                       double *out)
{
    double g = par[0], mc = par[1], mp = par[2], len = par[3], tau = par[4];
    double M = mp + mc, pml = mp * len;
    for (int b = 0; b < B; ++b) {
        const double *s4 = st + (size_t)b * 4;
        double x = s4[0], xd = s4[1], th = s4[2], thd = s4[3], force = u[b];
        double sn, cs;
        trig_sincos(math_mode, th, &sn, &cs);
        double temp = (force + pml * thd * thd * sn) / M;
        double thacc = (g * sn - cs * temp) / (len * (4.0 / 3.0 - mp * cs * cs / M));
        double xacc = temp - pml * thacc * cs / M;
        out[(size_t)b * 4 + 0] = x + tau * xd;
        out[(size_t)b * 4 + 1] = xd + tau * xacc;
        out[(size_t)b * 4 + 2] = th + tau * thd;
        out[(size_t)b * 4 + 3] = thd + tau * thacc;
    }
}
