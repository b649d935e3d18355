// Device-side building blocks of the batched control-limited iLQR solve for the
// cart-pole model (n = 5, m = 1), one problem per lane, all matrices in
// registers.  Each function names the reference code it stands for.
//
// Arithmetic contract (this is what makes results reproducible to the bit on
// any IEEE machine): fp64 throughout; the compiler must not contract
// (-ffp-contract=off); sums that the reference hands to BLAS/LAPACK are
// left-to-right chains that start with the first product and continue with
// explicit fma(); triangular solves multiply by the reciprocal of the diagonal;
// scalar expressions of the reference (dynamics, cost totals, schedules,
// ratios) are plain unfused operations in source order.  Structural zeros and
// ones of the cart-pole Jacobian are skipped: x*1 and x+0 are exact, so only the
// sign of an exact zero can differ from a dense evaluation.
#pragma once
#include <hip/hip_runtime.h>

#include "ilqr_math.hpp"

namespace ilqr {

constexpr int NX = 5;

// ---- kernel-argument structs (filled on the host, passed by value) ---------

struct CartpoleP {
    double dt, mc, mp, l, g;
};

struct QRCostP {      // QRCost, ilqr/ilqr_cost.py:490-530, m = 1
    double Q[25];     // Q
    double Qt[25];    // Q_terminal
    double QQ[25];    // Q + Q^T            (:526)
    double QQt[25];   // Q_terminal + Q_terminal^T (:528)
    double R, RR;     // R, R + R^T         (:527)
    double xg[5], ug; // x_goal, u_goal
};

struct QPOptsP { // ilqr_core.py:154-155
    int max_iter;
    int n_exhaust; // number of step reductions after which step < min_step (host-simulated)
    double min_grad, min_rel, step_dec, min_step, armijo;
};

struct SolveOptsP {
    int max_iter, n_alpha;
    double tol, scale, mu_min, mu_max, delta0;
    double alphas[32];
    QPOptsP qp;
};

// ---- cart-pole dynamics -----------------------------------------------------

struct CartpoleC { // constants derived once per thread
    double dt, g, l, mp, M, mpl, invM, mplM;
};

__device__ __forceinline__ CartpoleC cartpole_consts(const CartpoleP& p)
{
    CartpoleC c;
    c.dt = p.dt;
    c.g = p.g;
    c.l = p.l;
    c.mp = p.mp;
    c.M = p.mc + p.mp;
    c.mpl = p.mp * p.l;
    c.invM = 1.0 / c.M;
    c.mplM = c.mpl * c.invM;
    return c;
}

// CartPoleDynamics expression, ilqr/cartpole.py:72-92; x = [x, x', sin, cos, theta'].
__device__ __forceinline__ void cartpole_f(const CartpoleC& P, const double (&x)[NX], double F,
                                           double (&out)[NX])
{
    double xd = x[1], s = x[2], c = x[3], thd = x[4];
    double temp = (F + P.mpl * (thd * thd) * s) / P.M;
    double num = P.g * s - c * temp;
    double den = P.l * (4.0 / 3.0 - P.mp * (c * c) / P.M);
    double thdd = num / den;
    double xdd = temp - P.mpl * thdd * c / P.M;
    double th = det_atan2(s, c);
    double nth = th + thd * P.dt;
    double sn, cn;
    det_sincos(nth, sn, cn);
    out[0] = x[0] + xd * P.dt;
    out[1] = xd + xdd * P.dt;
    out[2] = sn;
    out[3] = cn;
    out[4] = thd + thdd * P.dt;
}

// Non-trivial entries of f_x (rows 1..4, columns 2..4) and f_u (rows 1, 4).
//   f_x = [1 dt 0   0   0  ]   f_u = [0 ]
//         [0 1  a12 a13 a14]         [b1]
//         [0 0  a22 a23 a24]         [0 ]
//         [0 0  a32 a33 a34]         [0 ]
//         [0 0  a42 a43 a44]         [b4]
struct CartpoleJac {
    double a[4][3]; // a[r-1][c-2]
    double b1, b4;
};

// Jacobian of the expression above with sin/cos as independent inputs (what
// T.grad yields, ilqr/diff_utils.py:28-37).  sn, cn = sin/cos of next_theta,
// i.e. entries 2, 3 of f(x, u).
__device__ __forceinline__ void cartpole_jac(const CartpoleC& P, const double (&x)[NX], double F,
                                             double sn, double cn, CartpoleJac& J)
{
    double s = x[2], c = x[3], thd = x[4];
    double invM = P.invM, mpl = P.mpl, dt = P.dt;
    double temp = (F + mpl * (thd * thd) * s) * invM;
    double num = P.g * s - c * temp;
    double den = P.l * (4.0 / 3.0 - P.mp * (c * c) * invM);
    double inv_den = 1.0 / den;
    double thdd = num * inv_den;
    double temp_s = mpl * (thd * thd) * invM;
    double temp_thd = 2.0 * mpl * thd * s * invM;
    double temp_F = invM;
    double num_s = P.g - c * temp_s;
    double num_c = -temp;
    double num_thd = -(c * temp_thd);
    double num_F = -(c * temp_F);
    double den_c = -(2.0 * P.l * P.mp * c * invM);
    double thdd_s = num_s * inv_den;
    double thdd_c = (num_c - thdd * den_c) * inv_den;
    double thdd_thd = num_thd * inv_den;
    double thdd_F = num_F * inv_den;
    double mplM = P.mplM;
    double xdd_s = temp_s - mplM * (thdd_s * c);
    double xdd_c = -(mplM * (thdd_c * c + thdd));
    double xdd_thd = temp_thd - mplM * (thdd_thd * c);
    double xdd_F = temp_F - mplM * (thdd_F * c);
    double r2 = s * s + c * c;
    double inv_r2 = 1.0 / r2;
    double th_s = c * inv_r2;
    double th_c = -(s * inv_r2);
    J.a[0][0] = xdd_s * dt;
    J.a[0][1] = xdd_c * dt;
    J.a[0][2] = xdd_thd * dt;
    J.a[1][0] = cn * th_s;
    J.a[1][1] = cn * th_c;
    J.a[1][2] = cn * dt;
    J.a[2][0] = -(sn * th_s);
    J.a[2][1] = -(sn * th_c);
    J.a[2][2] = -(sn * dt);
    J.a[3][0] = thdd_s * dt;
    J.a[3][1] = thdd_c * dt;
    J.a[3][2] = 1.0 + thdd_thd * dt;
    J.b1 = xdd_F * dt;
    J.b4 = thdd_F * dt;
}

// ---- QRCost -----------------------------------------------------------------

// (v^T M) . v in the order of x_diff.T.dot(Q).dot(x_diff), ilqr_cost.py:547
__device__ __forceinline__ double quad5(const double (&v)[NX], const double* M)
{
    double tot = 0.0;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        double acc = v[0] * M[j];
#pragma unroll
        for (int i = 1; i < NX; ++i)
            acc = fma(v[i], M[i * NX + j], acc);
        tot = (j == 0) ? acc * v[0] : fma(acc, v[j], tot);
    }
    return tot;
}

// QRCost.l, ilqr_cost.py:532-553 (the value carries the 1/2 its gradients lack)
__device__ __forceinline__ double qr_l(const QRCostP& C, const double (&x)[NX], double u)
{
    double xd[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
        xd[i] = x[i] - C.xg[i];
    double sx = quad5(xd, C.Q) * 0.5;
    double ud = u - C.ug;
    double su = ((ud * C.R) * ud) * 0.5;
    return sx + su;
}

__device__ __forceinline__ double qr_l_terminal(const QRCostP& C, const double (&x)[NX])
{
    double xd[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
        xd[i] = x[i] - C.xg[i];
    return quad5(xd, C.Qt) * 0.5;
}

// QRCost.l_x = x_diff^T (Q + Q^T), ilqr_cost.py:555-569
__device__ __forceinline__ void qr_lx(const double* QQ, const double* xg, const double (&x)[NX],
                                      double (&out)[NX])
{
    double xd[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i)
        xd[i] = x[i] - xg[i];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        double acc = xd[0] * QQ[j];
#pragma unroll
        for (int a = 1; a < NX; ++a)
            acc = fma(xd[a], QQ[a * NX + j], acc);
        out[j] = acc;
    }
}

// ---- boxQP, m = 1 -------------------------------------------------------------

__device__ __forceinline__ double clamp1(double lo, double hi, double v)
{
    // np.maximum(lower, np.minimum(upper, x)), NaN propagating (ilqr_boxQP.py:37)
    double t = (v != v) ? v : (hi < v ? hi : v);
    return (t != t) ? t : (lo > t ? lo : t);
}

__device__ __forceinline__ double qp_value1(double H, double g, double x)
{
    // np.dot(x.T, g) + 0.5 * np.dot(np.dot(x.T, H), x), ilqr_boxQP.py:62
    return x * g + 0.5 * ((x * H) * x);
}

struct QP1Out {
    double x;      // solution
    double rdiag;  // reciprocal of the 1x1 free-subspace Cholesky factor (1.0: never factorised, H_ff = ones)
    double hff;    // the factor itself (1.0 if never factorised)
    int result;    // codes of ilqr_boxQP.py:173-180
    bool free_;    // free mask after the last classification
    int n_iter, n_factor, n_backtrack;
};

// boxQP for one scalar unknown (ilqr_boxQP.py:9-186) with x0 given.  Every branch
// of the reference is kept, including the first pass that runs with the
// un-factorised H_ff = ones (:36, :108-117).
__device__ __forceinline__ void boxqp1(const QPOptsP& O, double H, double g, double lo, double hi,
                                       double x0, QP1Out& out)
{
    bool clamped = false, free_ = false;
    double oldvalue = 0.0;
    int result = -1;
    double rdiag = 1.0, hff = 1.0;
    int n_iter = 0, n_factor = 0, n_bt = 0;
    double x = clamp1(lo, hi, x0);
    double value = qp_value1(H, g, x);
    bool h_pd = H > 0.0; // np.all(np.linalg.eigvals(H) > 0) for a 1x1 matrix

    for (int iter = 0; iter < O.max_iter; ++iter) {
        if (result != -1)
            break;
        if (iter > 1 && (oldvalue - value) < O.min_rel * fabs(oldvalue) && (oldvalue - value) > 0.0) {
            result = 4;
            break;
        }
        oldvalue = value;
        double grad = g + H * x;
        bool oldclamped = clamped;
        clamped = ((x == lo) && (grad > 0.0)) || ((x == hi) && (grad < 0.0));
        free_ = !clamped;
        if (clamped) {
            result = 6;
            break;
        }
        if (!h_pd) {
            result = -1;
            break;
        }
        bool factorize = (iter == 1) || (oldclamped != clamped);
        if (factorize) {
            hff = sqrt(H);
            rdiag = 1.0 / hff;
            ++n_factor;
        }
        double gnorm = fabs(grad);
        if (gnorm < O.min_grad) {
            result = 5;
            break;
        }
        ++n_iter;
        double rhs = g + 0.0;
        double sol = (rhs * rdiag) * rdiag;
        double search = sol * -1 - x;

        double step = 1.0;
        int nstep = 0;
        double xc = clamp1(lo, hi, x + step * search);
        double sdotg = (xc - x) * grad;
        double vc = qp_value1(H, g, xc);
        while ((vc - oldvalue) / (sdotg + 1e-10) < O.armijo) {
            if (xc == x) {
                // Fixed point: x + step*search clamps (or rounds) back to x, and does so
                // for every smaller step, so vc == oldvalue and sdotg == 0 from here on.
                // The reference spins until step < minStep (:151-159); the number of
                // reductions that takes is a constant of (stepDec, minStep), computed
                // on the host with the same rounded multiplications.
                nstep = O.n_exhaust;
                result = 2;
                break;
            }
            step = step * O.step_dec;
            ++nstep;
            xc = clamp1(lo, hi, x + step * search);
            vc = qp_value1(H, g, xc);
            sdotg = (xc - x) * grad;
            if (step < O.min_step) {
                result = 2;
                break;
            }
        }
        n_bt += nstep;
        x = xc;
        value = vc;
    }
    out.x = x;
    out.rdiag = rdiag;
    out.hff = hff;
    out.result = result;
    out.free_ = free_;
    out.n_iter = n_iter;
    out.n_factor = n_factor;
    out.n_backtrack = n_bt;
}

} // namespace ilqr
