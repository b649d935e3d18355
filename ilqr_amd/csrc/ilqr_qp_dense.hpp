// boxQP (ilqr/ilqr_boxQP.py:9-186) for up to 8 unknowns as a per-lane scalar program: the
// literal algorithm with the reference's compaction of the free variables (index lists),
// used by the stand-alone kernel and by solvers whose control dimension is not 1.
// Sums the reference hands to BLAS are left-to-right chains (first product, then fma);
// triangular solves multiply by the reciprocal of the diagonal.  -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include "ilqr_device.hpp"

namespace ilqr {

constexpr int QM = 8; // largest boxQP dimension

// (v^T M) . v, ilqr_boxQP.py:62
__device__ inline double qpg_quad(int m, const double* v, const double* M)
{
    double tot = 0.0;
    for (int j = 0; j < m; ++j) {
        double acc = v[0] * M[j];
        for (int i = 1; i < m; ++i)
            acc = fma(v[i], M[i * m + j], acc);
        tot = (j == 0) ? acc * v[0] : fma(acc, v[j], tot);
    }
    return tot;
}

__device__ inline double qpg_value(int m, const double* H, const double* g, const double* x)
{
    double xg = x[0] * g[0];
    for (int i = 1; i < m; ++i)
        xg = fma(x[i], g[i], xg);
    return xg + 0.5 * qpg_quad(m, x, H);
}

// Lower Cholesky of the leading d x d block (row strides lda, ldl); 0 ok, 1 = pivot not > 0.
__device__ inline int qpg_chol(int d, const double* A, int lda, double* L, int ldl)
{
    for (int j = 0; j < d; ++j) {
        double s = A[j * lda + j];
        for (int k = 0; k < j; ++k)
            s = fma(-L[j * ldl + k], L[j * ldl + k], s);
        if (!(s > 0.0))
            return 1;
        double d_jj = sqrt(s);
        double r_jj = 1.0 / d_jj;
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

// Solve (L L^T) x = b in place, multiplying by reciprocal diagonals.
__device__ inline void qpg_solve(int d, const double* L, int ldl, double* b)
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

struct QPDenseOut {
    int result;   // codes of ilqr_boxQP.py:173-180
    int hd;       // dimension of the free-subspace factor in Hff (m if never factorised)
    int n_iter, n_factor, n_bt;
    bool linalg;  // the reference would raise LinAlgError (non-finite H, failed factorisation)
};

// H [m*m], g, lower, upper [m]; x [m] holds the start point on entry if have_x0, the solution on exit;
// Hff [m*m] (row stride m): the free-subspace factor, all ones until the first factorisation (:36);
// freev [m]: free mask of the last classification.
__device__ inline void boxqp_dense(const QPOptsP& O, int m, const double* H, const double* g, const double* lower,
                                   const double* upper, bool have_x0, double* x, double* Hff, int* freev,
                                   QPDenseOut& out)
{
    double grad[QM], search[QM], xc[QM], rhs[QM], sub[QM * QM], Lfull[QM * QM];
    int clamped[QM], oldclamped[QM], fidx[QM], cidx[QM];
    bool h_finite = true;
    for (int i = 0; i < m * m; ++i) {
        Hff[i] = 1.0; // H_ff = np.ones((n, n)), :36
        h_finite = h_finite && (fabs(H[i]) < __builtin_inf());
    }
    for (int i = 0; i < m; ++i) {
        clamped[i] = 0;
        freev[i] = 0;
        x[i] = have_x0 ? clamp1(lower[i], upper[i], x[i]) : (lower[i] + upper[i]) / 2;
    }
    double value = qpg_value(m, H, g, x), oldvalue = 0.0;
    int result = -1, hd = m, n_iter = 0, n_factor = 0, n_bt = 0;
    bool h_pd = h_finite && qpg_chol(m, H, m, Lfull, m) == 0;
    bool linalg = false;

    for (int iter = 0; iter < O.max_iter; ++iter) {
        if (result != -1)
            break;
        if (iter > 1 && (oldvalue - value) < O.min_rel * fabs(oldvalue) && (oldvalue - value) > 0.0) {
            result = 4;
            break;
        }
        oldvalue = value;
        for (int i = 0; i < m; ++i) {
            double acc = H[i * m] * x[0];
            for (int j = 1; j < m; ++j)
                acc = fma(H[i * m + j], x[j], acc);
            grad[i] = g[i] + acc;
        }
        int nclamped = 0;
        for (int i = 0; i < m; ++i) {
            oldclamped[i] = clamped[i];
            clamped[i] = ((x[i] == lower[i]) && (grad[i] > 0.0)) || ((x[i] == upper[i]) && (grad[i] < 0.0));
            freev[i] = !clamped[i];
            nclamped += clamped[i];
        }
        if (nclamped == m) {
            result = 6;
            break;
        }
        if (!h_finite) { // eigvals raises LinAlgError
            linalg = true;
            break;
        }
        if (!h_pd) {
            result = -1;
            break;
        }
        bool factorize = iter == 1;
        for (int i = 0; i < m; ++i)
            factorize = factorize || (oldclamped[i] != clamped[i]);
        int nf = 0, nc = 0;
        for (int i = 0; i < m; ++i) {
            if (freev[i])
                fidx[nf++] = i;
            else
                cidx[nc++] = i;
        }
        if (factorize) {
            for (int a = 0; a < nf; ++a)
                for (int b = 0; b < nf; ++b)
                    sub[a * m + b] = H[fidx[a] * m + fidx[b]];
            if (qpg_chol(nf, sub, m, Hff, m) != 0) {
                linalg = true;
                break;
            }
            hd = nf;
            ++n_factor;
        }
        double ss = 0.0;
        for (int a = 0; a < nf; ++a)
            ss = (a == 0) ? grad[fidx[0]] * grad[fidx[0]] : fma(grad[fidx[a]], grad[fidx[a]], ss);
        double gnorm = nf == 1 ? fabs(grad[fidx[0]]) : sqrt(ss);
        if (gnorm < O.min_grad) {
            result = 5;
            break;
        }
        ++n_iter;
        for (int a = 0; a < nf; ++a) {
            double acc = 0.0;
            for (int b = 0; b < nc; ++b)
                acc = (b == 0) ? H[fidx[a] * m + cidx[0]] * x[cidx[0]]
                               : fma(H[fidx[a] * m + cidx[b]], x[cidx[b]], acc);
            rhs[a] = g[fidx[a]] + acc;
        }
        qpg_solve(nf, Hff, m, rhs);
        for (int i = 0; i < m; ++i)
            search[i] = 0.0;
        for (int a = 0; a < nf; ++a)
            search[fidx[a]] = rhs[a] * -1 - x[fidx[a]];

        double step = 1.0;
        for (int i = 0; i < m; ++i)
            xc[i] = clamp1(lower[i], upper[i], x[i] + step * search[i]);
        double sdotg = (xc[0] - x[0]) * grad[0];
        for (int i = 1; i < m; ++i)
            sdotg = sdotg + (xc[i] - x[i]) * grad[i];
        double vc = qpg_value(m, H, g, xc);
        while ((vc - oldvalue) / (sdotg + 1e-10) < O.armijo) {
            step = step * O.step_dec;
            ++n_bt;
            for (int i = 0; i < m; ++i)
                xc[i] = clamp1(lower[i], upper[i], x[i] + step * search[i]);
            vc = qpg_value(m, H, g, xc);
            sdotg = (xc[0] - x[0]) * grad[0];
            for (int i = 1; i < m; ++i)
                sdotg = sdotg + (xc[i] - x[i]) * grad[i];
            if (step < O.min_step) {
                result = 2;
                break;
            }
        }
        for (int i = 0; i < m; ++i)
            x[i] = xc[i];
        value = vc;
    }
    out.result = result;
    out.hd = hd;
    out.n_iter = n_iter;
    out.n_factor = n_factor;
    out.n_bt = n_bt;
    out.linalg = linalg;
}

} // namespace ilqr
