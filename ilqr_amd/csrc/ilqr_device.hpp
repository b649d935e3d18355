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
// ones (of the cart-pole Jacobian, and of a diagonal cost matrix) are skipped:
// x*1 and x+0 are exact, so only the sign of an exact zero can differ from a
// dense evaluation.
#pragma once
#include <hip/hip_runtime.h>

#include "ilqr_math.hpp"

namespace ilqr {

constexpr int NX = 5;

// Pin a wave-uniform value in vector registers.  Kernel arguments live in scalar
// registers; the cost matrices, model constants and options together exceed the
// ~100 SGPRs a wave has, and the compiler then spills them lane by lane
// (v_writelane / v_readlane around every use).  There are VGPRs to spare.
__device__ __forceinline__ double vreg(double x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// Keep a value's registers occupied up to this point.  A global store holds its
// data registers until it completes (~2 us); if the allocator hands them to the
// next computation, the hardware-required wait for the store lands in the middle
// of the step.  Results are therefore parked in "pending" variables that are
// stored at the START of the next step and kept alive (this anchor) to its END,
// so that by the time they are overwritten the store has long retired.
__device__ __forceinline__ void keep_alive(double x) { asm volatile("" ::"v"(x)); }

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
    // PathQRCost (ilqr/ilqr_cost.py:635-770): goals indexed by the time step, shared by the
    // batch; device arrays [(T+1)*5] and [T], null for QRCost
    const double* x_path;
    const double* u_path;
    // Per-problem goals (PathQR evaluator only): the goals travel in the solve's workspace like the trajectories
    // (rows of a batch-minor array, see PathQR::bind); goal_t_x / goal_t_u = rows per time step (5 and 1 for
    // per-problem paths, 0 for per-problem goals that do not depend on the step); vscale = 0.5 gives the values
    // QRCost's factor (ilqr_cost.py:547, :553), 1.0 is PathQRCost's (:695, :701).
    int goal_t_x, goal_t_u;
    double vscale;
};

struct QPOptsP { // ilqr_core.py:154-155
    int max_iter;
    int n_exhaust;    // number of step reductions after which step < min_step (host-simulated)
    double step_last; // the step after n_exhaust reductions
    double min_grad, min_rel, step_dec, min_step, armijo;
    int tab_n;          // entries of the step table the Armijo search may jump to: min(n_exhaust, QP_TAB)
    float inv_log2_dec; // 1 / log2(step_dec), for the (unverified, fp32) estimate of the jump target
};

// The Armijo steps 1, dec, dec*dec, ... exactly as the loop's repeated rounded
// multiplication produces them (ilqr_boxQP.py:151), so that the search can resume
// at trip t with the very bits the literal loop would hold there.  One table per
// workgroup, in LDS, filled once at kernel start.
constexpr int QP_TAB = 512;

__device__ __forceinline__ void qp_fill_step_table(double* tab, double step_dec)
{
    if (threadIdx.x == 0) {
        double s = 1.0;
        for (int t = 0; t < QP_TAB; ++t) {
            tab[t] = s;
            s = s * step_dec;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ QPOptsP qp_opts_in_vgprs(const QPOptsP& o)
{
    QPOptsP v = o;
    v.step_last = vreg(o.step_last);
    v.min_grad = vreg(o.min_grad);
    v.min_rel = vreg(o.min_rel);
    v.step_dec = vreg(o.step_dec);
    v.min_step = vreg(o.min_step);
    v.armijo = vreg(o.armijo);
    return v;
}

struct SolveOptsP {
    int max_iter, n_alpha;
    int mu_from_array; // 0: mu starts at 1.0 (ilqr_core.py:264); 1: at mu_out[b]
    int lanes_per_wave; // problems handled by one wavefront of the solve kernel (64, 32 or 16)
    double tol, scale, mu_min, mu_max, delta0;
    double alphas[32];
    QPOptsP qp;
    double* dJ_carry; // [2, B] or null: Delta_J1 / Delta_J2 carried from one launch to the next (ilqr_core.py:278-279, :305)
};

// ---- cart-pole dynamics -----------------------------------------------------

struct CartpoleC { // constants derived once per thread, held in VGPRs
    double dt, g, l, mp, M, mpl, invM, mplM;
};

__device__ __forceinline__ CartpoleC cartpole_consts(const CartpoleP& p)
{
    CartpoleC c;
    c.dt = vreg(p.dt);
    c.g = vreg(p.g);
    c.l = vreg(p.l);
    c.mp = vreg(p.mp);
    c.M = vreg(p.mc + p.mp);
    c.mpl = vreg(p.mp * p.l);
    c.invM = vreg(1.0 / c.M);
    c.mplM = vreg(c.mpl * c.invM);
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

// The same expression in two independent halves (a step's next velocities do not depend on its next angle
// and vice versa), for a rollout shared by two wavefronts: same operations, same bits.
__device__ __forceinline__ void cartpole_f_velocities(const CartpoleC& P, const double (&x)[NX], double F,
                                                      double& out0, double& out1, double& out4)
{
    double xd = x[1], s = x[2], c = x[3], thd = x[4];
    double temp = (F + P.mpl * (thd * thd) * s) / P.M;
    double num = P.g * s - c * temp;
    double den = P.l * (4.0 / 3.0 - P.mp * (c * c) / P.M);
    double thdd = num / den;
    double xdd = temp - P.mpl * thdd * c / P.M;
    out0 = x[0] + xd * P.dt;
    out1 = xd + xdd * P.dt;
    out4 = thd + thdd * P.dt;
}
__device__ __forceinline__ void cartpole_f_angle(const CartpoleC& P, double s, double c, double thd, double& sn,
                                                 double& cn)
{
    double th = det_atan2(s, c);
    double nth = th + thd * P.dt;
    det_sincos(nth, sn, cn);
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

// ---- QRCost (ilqr/ilqr_cost.py:486-632), two evaluators with one interface ----
//
//   l(x, u), l_terminal(x)        :532-553  (value carries the 1/2 its gradients lack)
//   lx(x, out), lx_terminal       :555-569  x_diff^T (Q + Q^T)
//   lu(u)                         :571-587
//   lxx(a, b), lxx_terminal(a, b) :589-601
//   add_lxx(a, b, v) = lxx(a, b) + v
//   luu()                         :617-632
//   at(i), at_terminal(N)         select the time step (only PathQR depends on it)
//
// DenseQR reads the full matrices from the kernel arguments.  DiagQR is used when
// Q and Q_terminal are diagonal (the host checks): it keeps the ten diagonal
// entries in VGPRs and evaluates exactly the non-zero terms of the dense chains.

struct DenseQR {
    const QRCostP& C;
    __device__ __forceinline__ explicit DenseQR(const QRCostP& c) : C(c) {}
    __device__ __forceinline__ void at(int) const {}          // the goal does not depend on the step
    __device__ __forceinline__ void at_terminal(int) const {}
    __device__ __forceinline__ void bind(const double*, const double*, int64_t) const {} // ... nor on the problem
    __device__ __forceinline__ void column(int64_t) const {}

    __device__ __forceinline__ double quad(const double (&v)[NX], const double* M) const
    {
        // (v^T M) . v in the order of x_diff.T.dot(Q).dot(x_diff), :547
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
    __device__ __forceinline__ void diff(const double (&x)[NX], double (&xd)[NX]) const
    {
#pragma unroll
        for (int i = 0; i < NX; ++i)
            xd[i] = x[i] - C.xg[i];
    }
    __device__ __forceinline__ double l(const double (&x)[NX], double u) const
    {
        double xd[NX];
        diff(x, xd);
        double sx = quad(xd, C.Q) * 0.5;
        double ud = u - C.ug;
        double su = ((ud * C.R) * ud) * 0.5;
        return sx + su;
    }
    __device__ __forceinline__ double l_terminal(const double (&x)[NX]) const
    {
        double xd[NX];
        diff(x, xd);
        return quad(xd, C.Qt) * 0.5;
    }
    __device__ __forceinline__ void lx_with(const double* QQ, const double (&x)[NX], double (&out)[NX]) const
    {
        double xd[NX];
        diff(x, xd);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            double acc = xd[0] * QQ[j];
#pragma unroll
            for (int a = 1; a < NX; ++a)
                acc = fma(xd[a], QQ[a * NX + j], acc);
            out[j] = acc;
        }
    }
    __device__ __forceinline__ void lx(const double (&x)[NX], double (&out)[NX]) const { lx_with(C.QQ, x, out); }
    __device__ __forceinline__ void lx_terminal(const double (&x)[NX], double (&out)[NX]) const
    {
        lx_with(C.QQt, x, out);
    }
    __device__ __forceinline__ double lu(double u) const { return (u - C.ug) * C.RR; }
    __device__ __forceinline__ double lxx(int a, int b) const { return C.QQ[a * NX + b]; }
    __device__ __forceinline__ double lxx_terminal(int a, int b) const { return C.QQt[a * NX + b]; }
    __device__ __forceinline__ double add_lxx(int a, int b, double v) const { return C.QQ[a * NX + b] + v; }
    __device__ __forceinline__ double luu() const { return C.RR; }
};

struct DiagQR {
    double q[NX], qt[NX], qq[NX], qqt[NX], xg[NX], R, RR, ug;
    __device__ __forceinline__ explicit DiagQR(const QRCostP& C)
    {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            q[i] = vreg(C.Q[i * NX + i]);
            qt[i] = vreg(C.Qt[i * NX + i]);
            qq[i] = vreg(C.QQ[i * NX + i]);
            qqt[i] = vreg(C.QQt[i * NX + i]);
            xg[i] = vreg(C.xg[i]);
        }
        R = vreg(C.R);
        RR = vreg(C.RR);
        ug = vreg(C.ug);
    }
    __device__ __forceinline__ void at(int) const {}
    __device__ __forceinline__ void at_terminal(int) const {}
    __device__ __forceinline__ void bind(const double*, const double*, int64_t) const {}
    __device__ __forceinline__ void column(int64_t) const {}
    __device__ __forceinline__ double quad(const double (&v)[NX], const double (&d)[NX]) const
    {
        // the dense chain with every off-diagonal product dropped: column j of
        // v^T M is v[j] * d[j], then the running dot with v
        double tot = (v[0] * d[0]) * v[0];
#pragma unroll
        for (int j = 1; j < NX; ++j)
            tot = fma(v[j] * d[j], v[j], tot);
        return tot;
    }
    __device__ __forceinline__ void diff(const double (&x)[NX], double (&xd)[NX]) const
    {
#pragma unroll
        for (int i = 0; i < NX; ++i)
            xd[i] = x[i] - xg[i];
    }
    __device__ __forceinline__ double l(const double (&x)[NX], double u) const
    {
        double xd[NX];
        diff(x, xd);
        double sx = quad(xd, q) * 0.5;
        double ud = u - ug;
        double su = ((ud * R) * ud) * 0.5;
        return sx + su;
    }
    __device__ __forceinline__ double l_terminal(const double (&x)[NX]) const
    {
        double xd[NX];
        diff(x, xd);
        return quad(xd, qt) * 0.5;
    }
    __device__ __forceinline__ void lx(const double (&x)[NX], double (&out)[NX]) const
    {
#pragma unroll
        for (int j = 0; j < NX; ++j)
            out[j] = (x[j] - xg[j]) * qq[j];
    }
    __device__ __forceinline__ void lx_terminal(const double (&x)[NX], double (&out)[NX]) const
    {
#pragma unroll
        for (int j = 0; j < NX; ++j)
            out[j] = (x[j] - xg[j]) * qqt[j];
    }
    __device__ __forceinline__ double lu(double u) const { return (u - ug) * RR; }
    __device__ __forceinline__ double lxx(int a, int b) const { return a == b ? qq[a] : 0.0; }
    __device__ __forceinline__ double lxx_terminal(int a, int b) const { return a == b ? qqt[a] : 0.0; }
    __device__ __forceinline__ double add_lxx(int a, int b, double v) const { return a == b ? qq[a] + v : v; }
    __device__ __forceinline__ double luu() const { return RR; }
};


// PathQRCost (ilqr/ilqr_cost.py:635-770): the goals are x_path[i], u_path[i], and the value
// has NO 1/2 (:695, :701) while the gradients and Hessians are those of QRCost.  at(i) /
// at_terminal(N) select the step before any evaluation.
// The same evaluator serves goals that differ from problem to problem (every problem of the batch its own
// QRCost goal or its own PathQRCost path, as B separate solver objects of the reference would have): bind()
// hands it the goal rows of the wavefront's workspace tile, column() selects the problem whose trajectory is
// being evaluated (not always the lane's own: the cooperative line search rolls out other lanes' problems),
// and C.vscale restores QRCost's 1/2 on the values.  Without bind() the path is the batch's shared one
// (uniform loads).
struct PathQR {
    const QRCostP& C;
    mutable double xg[NX], ug;
    mutable const double *gx, *gu; // null: shared path
    mutable int64_t gld, col;
    __device__ __forceinline__ explicit PathQR(const QRCostP& c) : C(c), gx(nullptr), gu(nullptr), gld(0), col(0)
    {
#pragma unroll
        for (int i = 0; i < NX; ++i)
            xg[i] = 0.0;
        ug = 0.0;
    }
    __device__ __forceinline__ void bind(const double* goal_x, const double* goal_u, int64_t ld) const
    {
        gx = goal_x;
        gu = goal_u;
        gld = ld;
    }
    __device__ __forceinline__ void column(int64_t b) const { col = b; }
    __device__ __forceinline__ void at(int t) const
    {
        if (gx != nullptr) {
#pragma unroll
            for (int i = 0; i < NX; ++i)
                xg[i] = gx[((int64_t)t * C.goal_t_x + i) * gld + col];
            ug = gu[(int64_t)t * C.goal_t_u * gld + col];
            return;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i)
            xg[i] = C.x_path[t * NX + i];
        ug = C.u_path[t];
    }
    __device__ __forceinline__ void at_terminal(int T) const
    {
        if (gx != nullptr) {
#pragma unroll
            for (int i = 0; i < NX; ++i)
                xg[i] = gx[((int64_t)T * C.goal_t_x + i) * gld + col];
            return;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i)
            xg[i] = C.x_path[T * NX + i];
    }
    __device__ __forceinline__ double quad(const double (&v)[NX], const double* M) const
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
    __device__ __forceinline__ void diff(const double (&x)[NX], double (&xd)[NX]) const
    {
#pragma unroll
        for (int i = 0; i < NX; ++i)
            xd[i] = x[i] - xg[i];
    }
    __device__ __forceinline__ double l(const double (&x)[NX], double u) const
    {
        double xd[NX];
        diff(x, xd);
        double ud = u - ug;
        return quad(xd, C.Q) * C.vscale + ((ud * C.R) * ud) * C.vscale; // (x * 1.0 is x: PathQRCost's bits)
    }
    __device__ __forceinline__ double l_terminal(const double (&x)[NX]) const
    {
        double xd[NX];
        diff(x, xd);
        return quad(xd, C.Qt) * C.vscale;
    }
    __device__ __forceinline__ void lx_with(const double* QQ, const double (&x)[NX], double (&out)[NX]) const
    {
        double xd[NX];
        diff(x, xd);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            double acc = xd[0] * QQ[j];
#pragma unroll
            for (int a = 1; a < NX; ++a)
                acc = fma(xd[a], QQ[a * NX + j], acc);
            out[j] = acc;
        }
    }
    __device__ __forceinline__ void lx(const double (&x)[NX], double (&out)[NX]) const { lx_with(C.QQ, x, out); }
    __device__ __forceinline__ void lx_terminal(const double (&x)[NX], double (&out)[NX]) const
    {
        lx_with(C.QQt, x, out);
    }
    __device__ __forceinline__ double lu(double u) const { return (u - ug) * C.RR; }
    __device__ __forceinline__ double lxx(int a, int b) const { return C.QQ[a * NX + b]; }
    __device__ __forceinline__ double lxx_terminal(int a, int b) const { return C.QQt[a * NX + b]; }
    __device__ __forceinline__ double add_lxx(int a, int b, double v) const { return C.QQ[a * NX + b] + v; }
    __device__ __forceinline__ double luu() const { return C.RR; }
};

// ---- boxQP, m = 1 -------------------------------------------------------------

__device__ __forceinline__ double clamp1(double lo, double hi, double v)
{
    // np.maximum(lower, np.minimum(upper, x)), NaN propagating (ilqr_boxQP.py:37).
    // fmin/fmax return the other operand when one is NaN, so NaNs (never seen on a
    // healthy trajectory) are put back by hand behind a wave-uniform branch.
    double r = fmax(lo, fmin(hi, v));
    if (__ballot((v != v) | (lo != lo) | (hi != hi)) != 0ull) {
        double t = (v != v) ? v : (hi < v ? hi : v);
        r = (t != t) ? t : (lo > t ? lo : t);
    }
    return r;
}

__device__ __forceinline__ double qp_value1(double H, double g, double x)
{
    // np.dot(x.T, g) + 0.5 * np.dot(np.dot(x.T, H), x), ilqr_boxQP.py:62
    return x * g + 0.5 * ((x * H) * x);
}

// (num / den) < a, decided without the division whenever the outcome is not within
// rounding distance of the threshold: for den > 0, num < a*den*(1 - 1e-12) implies
// the correctly rounded quotient is below a, num > a*den*(1 + 1e-12) implies it is
// not (rounding is monotone and a is representable); signs flip for den < 0.
// Anything else (near ties, den == 0, NaN, inf, a <= 0) takes the real division,
// behind a wave-uniform branch so that the common path is straight-line code.
__device__ __forceinline__ bool quotient_below(double num, double den, double a)
{
    double p = a * den;
    double tol = fabs(p) * 1e-12;
    bool lo = num < p - tol, hi = num > p + tol;
    bool pos = den > 0.0, neg = den < 0.0;
    bool usable = a > 0.0; // a <= 0 would let an underflowing quotient change the answer
    bool sure_true = usable & ((pos & lo) | (neg & hi));
    bool sure_false = usable & ((pos & hi) | (neg & lo));
    bool below = sure_true;
    if (__ballot(!(sure_true | sure_false)) != 0ull) {
        bool exact = (num / den) < a;
        below = (sure_true | sure_false) ? sure_true : exact;
    }
    return below;
}

// ---- proofs used by the Armijo search of boxqp1 -------------------------------
//
// Notation.  x, grad, search, oldvalue are fixed during one search; a candidate is
// xc = clamp(x + step * search), d = xc - x (exact), u = |d|, eps = 1e-10, a = the
// Armijo constant in (0, 1), G = |computed grad|, gerr >= |computed - exact grad|.
// In exact arithmetic  value(xc) - value(x) = d * grad + H d^2 / 2.  The test of
// ilqr_boxQP.py:151 continues while  num / den < a  with
//   num = fl(value(xc)) - fl(value(x)),   den = fl(fl(d) * grad) + eps.
// |num - (d grad + H d^2 / 2)| <= 2e-15 (X |g| + H X^2) for |x|, |xc| <= X (rounding of
// the two quadratic evaluations and of their difference); den carries a relative
// error of a few ulp.  qp_error_budget() returns a bound Ep on every rounding term
// that enters the two proofs below (including the rounding of evaluating h itself),
// for candidates with u <= D.
//
// With d * grad = -u G (descent):  num - a den = h(u) + rounding,
//   h(u) = H u^2 / 2 - (1 - a) G u - a eps,   convex, h(0) < 0.
//  * den > 0 (u G < eps):  continue  <=>  num < a den  <=>  h(u) < 0.  If that holds at
//    u = D with margin Ep it holds on all of [0, D] (convexity), i.e. for every later,
//    shorter candidate: the loop can only end by step < min_step.   [spin]
//  * den < 0 (u G > eps):  continue  <=>  num > a den  <=>  h(u) > 0.  If that holds at
//    u = uv with margin Ep it holds for every u >= uv (h increases beyond its root),
//    i.e. for every EARLIER, longer candidate: u is monotone in step because rounding,
//    clamping and the products by fixed numbers are monotone.        [jump]
// Without a sure sign of d * grad the spin proof uses the bounds  num <= u G + H u^2 / 2,
// den >= eps - u G  instead.

__device__ __forceinline__ double qp_error_budget(const QPOptsP& O, double H, double g, double ax, double G,
                                                  double gerr, double D)
{
    double X = ax + D;
    return 2e-15 * (X * fabs(g) + (H * X) * X) + 1e-14 * ((0.5 * H * D) * D + D * G + O.armijo * 1e-10) +
           2.0 * (D * gerr);
}

// True only if every candidate with |xc - x| <= D fails the Armijo test (so the
// search ends at step < min_step with result 2, ilqr_boxQP.py:153-156).
__device__ __forceinline__ bool qp_spin_proved(const QPOptsP& O, double H, double G, double gerr, bool descent,
                                               double D, double Ep)
{
    double q = (0.5 * H * D) * D;
    double lin = descent ? -((1.0 - O.armijo) * (G - gerr)) : (1.0 + O.armijo) * (G + gerr);
    double hD = (q + lin * D) - O.armijo * 1e-10;
    bool den_pos = (D * (G + gerr)) * 1.000000001 < 0.999999999e-10;
    return den_pos & (hD + Ep < 0.0) & (Ep < O.armijo * 1e-10);
}

#ifdef ILQR_QP_PROF // diagnostic build only: per-wave event counts of boxqp1
static __device__ unsigned long long g_qp_prof[16];
__device__ __forceinline__ void qp_count(int slot)
{
    if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1)
        atomicAdd(&g_qp_prof[slot], 1ull);
}
#define QP_COUNT(slot) qp_count(slot)
// per-wave counts (index: global wavefront number of a 1-D launch): [0] literal-loop fallbacks,
// [1] calls in which some lane's candidate was rejected
static __device__ unsigned int g_qp_wave[2][8192];
__device__ __forceinline__ void qp_count_wave(int slot)
{
    if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) {
        unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if (w < 8192u)
            atomicAdd(&g_qp_wave[slot][w], 1u);
    }
}
#define QP_COUNT_WAVE(slot) qp_count_wave(slot)
#else
#define QP_COUNT_WAVE(slot)
#define QP_COUNT(slot)
#endif

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
// sqrtH, rsqrtH: sqrt(H) and 1.0 / sqrt(H) as the caller already has them from
// its own Cholesky of the same 1x1 matrix (same input, same bits).
// tab: the workgroup's step table (qp_fill_step_table).
// This is the literal loop, any number of passes; boxqp1() below tries the common
// three-pass shape as straight-line code first.
__device__ __forceinline__ void boxqp1_loop(const QPOptsP& O, const double* tab, double H, double sqrtH, double rsqrtH,
                                            double g, double lo, double hi, double x0, QP1Out& out)
{
    bool clamped = false, free_ = false;
    double oldvalue = 0.0;
    int result = -1;
    double rdiag = 1.0, hff = 1.0;
    int n_iter = 0, n_factor = 0, n_bt = 0;
    double x = clamp1(lo, hi, x0);
    double value = qp_value1(H, g, x);
    bool h_pd = H > 0.0; // np.all(np.linalg.eigvals(H) > 0) for a 1x1 matrix

    // The main loop (:85-171) is written with a per-lane `active` flag and selects instead of
    // breaks: lanes leave it at different points (result 4, 6, 5, not PD, 2), and straight-line
    // predicated code costs far fewer scalar instructions than the exec-mask bookkeeping of
    // divergent branches.  A lane that has left keeps all its state; the loop ends when no lane
    // of the wave is active.
    bool active = true;
    double grad = 0.0;
    QP_COUNT(0);
    int passes = 0;
    for (int iter = 0; iter < O.max_iter; ++iter) {
        if (__ballot(active) == 0ull)
            break;
        QP_COUNT(1);
        ++passes;
        bool ex4 = active & (iter > 1) & ((oldvalue - value) < O.min_rel * fabs(oldvalue)) & ((oldvalue - value) > 0.0);
        result = ex4 ? 4 : result;
        active = active & !ex4;
        oldvalue = active ? value : oldvalue;
        grad = active ? g + H * x : grad;
        bool oldclamped = clamped;
        bool cl = ((x == lo) & (grad > 0.0)) | ((x == hi) & (grad < 0.0));
        clamped = active ? cl : clamped;
        free_ = active ? !cl : free_;
        bool ex6 = active & cl;
        result = ex6 ? 6 : result;
        active = active & !ex6 & h_pd; // not PD: leave with result -1 (:103-105)
        bool factorize = active & ((iter == 1) | (oldclamped != clamped));
        hff = factorize ? sqrtH : hff;
        rdiag = factorize ? rsqrtH : rdiag;
        n_factor += factorize ? 1 : 0;
        bool ex5 = active & (fabs(grad) < O.min_grad);
        result = ex5 ? 5 : result;
        active = active & !ex5;
        n_iter += active ? 1 : 0;
        if (__ballot(active) == 0ull) // typical last pass: every lane has just left
            break;
        QP_COUNT(2);
        double rhs = g + 0.0;
        double sol = (rhs * rdiag) * rdiag;
        double search = sol * -1 - x;

        // Armijo line search (:140-159), written with selects and a single loop exit.
        // The reference walks step = 1, dec, dec^2, ... and tests every candidate; a
        // typical first pass (un-factorised, H_ff = ones) rejects a dozen of them and a
        // converged problem spins through all n_exhaust.  Both are cut short here by
        // PROOFS about the outcome of the skipped tests, so that no result, counter or
        // bit changes (see qp_spin_proved and the jump below).
        double step = 1.0;
        int nstep = 0;
        double xc = clamp1(lo, hi, x + step * search);
        double sdotg = (xc - x) * grad;
        double vc = qp_value1(H, g, xc);
        bool more = active & quotient_below(vc - oldvalue, sdotg + 1e-10, O.armijo);
        if (__ballot(more) != 0ull) {
            QP_COUNT(3);
            if (iter >= 1)
                QP_COUNT(6);
            double ax = fabs(x), G = fabs(grad), as = fabs(search);
            double gerr = 5e-16 * (fabs(g) + fabs(H * x)); // |computed grad - exact g + H x| stays below this
            bool descent = (((search > 0.0) & (grad < 0.0)) | ((search < 0.0) & (grad > 0.0))) & (G > 4.0 * gerr);
            bool a_ok = (O.armijo > 0.0) & (O.armijo < 1.0);
            double D0 = 1.001 * as + 4e-16 * ax; // |xc' - x| <= D0 for this and every later candidate
            double Ep = qp_error_budget(O, H, g, ax, G, gerr, D0);
            bool exhaust = more & ((xc == x) | (a_ok & qp_spin_proved(O, H, G, gerr, descent, D0, Ep)));
            // Jump: the search may resume at trip tj if every trip before it provably
            // continues.  tj comes from an fp32 estimate of where the test flips (its
            // accuracy only decides how much is skipped); the proof is evaluated in fp64
            // at trip tj - 1 and covers all earlier trips by monotonicity.
            // (hardware reciprocal / square root / log2 approximations: the estimate needs no accuracy)
            float Gf = (float)G, Hf = (float)H, cf = (float)(1.0 - O.armijo);
            float cg = cf * Gf;
            float up = (cg + __builtin_amdgcn_sqrtf(cg * cg + (float)(2e-10 * O.armijo) * Hf)) * __builtin_amdgcn_rcpf(Hf);
            float U = fmaxf(up, 1e-10f * __builtin_amdgcn_rcpf(Gf));
            float tf = __builtin_amdgcn_logf(U * __builtin_amdgcn_rcpf((float)as)) * O.inv_log2_dec;
            tf = fminf(fmaxf(tf, 0.0f), 1.0e6f); // NaN -> 0
            int tj = (int)tf + 1;
            tj = tj < O.tab_n ? tj : O.tab_n - 1;
            bool try_jump = more & !exhaust & descent & a_ok & (tj >= 2);
            tj = try_jump ? tj : 1;
            double sv = tab[tj - 1], sj = tab[tj];
            double xv = clamp1(lo, hi, x + sv * search);
            double dv = xv - x, uv = fabs(dv);
            double hv = (0.5 * H * uv) * uv - ((1.0 - O.armijo) * (G + gerr)) * uv - O.armijo * 1e-10;
            bool jump = try_jump & (hv > Ep) & (fabs(dv * grad) > 1.0000001e-10);
            step = exhaust ? O.step_last : (jump ? sj : step);
            nstep = exhaust ? O.n_exhaust : (jump ? tj : nstep);
            result = exhaust ? 2 : result;
            if (__ballot(exhaust | jump) != 0ull) {
                QP_COUNT(4);
                double xn = clamp1(lo, hi, x + step * search);
                xc = (exhaust | jump) ? xn : xc;
                vc = qp_value1(H, g, xc);
                sdotg = (xc - x) * grad;
                bool m2 = quotient_below(vc - oldvalue, sdotg + 1e-10, O.armijo);
                more = exhaust ? false : (jump ? m2 : more);
            }
            int trip = 0; // same for every lane still in the loop
            while (more) {
                QP_COUNT(5);
                bool ex = (xc == x);
                if ((trip & 3) == 3) {
                    double D = 1.001 * (step * as) + 4e-16 * ax;
                    ex = ex | (a_ok & qp_spin_proved(O, H, G, gerr, descent, D, qp_error_budget(O, H, g, ax, G, gerr, D)));
                }
                step = ex ? O.step_last : step * O.step_dec;
                nstep = ex ? O.n_exhaust : nstep + 1;
                ++trip;
                xc = clamp1(lo, hi, x + step * search);
                vc = qp_value1(H, g, xc);
                sdotg = (xc - x) * grad;
                bool ended = ex | (step < O.min_step);
                result = ended ? 2 : result;
                more = quotient_below(vc - oldvalue, sdotg + 1e-10, O.armijo) & !ended;
            }
        }
        n_bt += active ? nstep : 0;
        x = active ? xc : x;
        value = active ? vc : value;
        active = active & (result == -1);
    }
    (void)passes;
    out.x = x;
    out.rdiag = rdiag;
    out.hff = hff;
    out.result = result;
    out.free_ = free_;
    out.n_iter = n_iter;
    out.n_factor = n_factor;
    out.n_backtrack = n_bt;
}

// ---- the common shape of boxqp1_loop as (almost) straight-line code ------------
//
// Almost every call makes exactly three passes: pass 0 (H_ff = ones) backtracks to some
// trip t_j and accepts, pass 1 (factorised) accepts the full Newton step or exhausts the
// search near convergence, pass 2 ends on one of the termination tests.  boxqp1() runs
// that shape for all lanes at once: no per-pass predication (a lane that has left simply
// computes values nobody reads; the outputs are selected at the end from the stage at
// which the lane left), one evaluated candidate per pass (pass 0: at tab[t_j] when the
// jump proof shows that every earlier trip continues, which covers the candidate at
// step 1 as well), and the exhaustion proof behind a wave-uniform branch that is taken
// only when some lane's candidate was rejected.  Same operations on the same operands as
// the loop, hence the same bits and counters.  A lane whose call has any other shape (a
// rejected candidate without proof, a fourth pass, a NaN, an undecided quotient) reports
// `slow`, and the wave then runs the literal loop from the start.

// fmax(lo, fmin(hi, v)); NaN operands are left to the literal loop.
__device__ __forceinline__ double qp_clamp_fast(double lo, double hi, double v, bool& slow)
{
    slow = slow | (v != v);
    return fmax(lo, fmin(hi, v));
}

// quotient_below without the division: undecided cases are left to the literal loop.
__device__ __forceinline__ bool qp_quotient_fast(double num, double den, double a, bool& slow)
{
    double p = a * den;
    double tol = fabs(p) * 1e-12;
    bool lo = num < p - tol, hi = num > p + tol;
    bool pos = den > 0.0, neg = den < 0.0;
    bool sure_true = (pos & lo) | (neg & hi);
    bool sure_false = (pos & hi) | (neg & lo);
    slow = slow | !(sure_true | sure_false);
    return sure_true;
}

// One candidate of the line search (ilqr_boxQP.py:141-151): true if the search goes on.
__device__ __forceinline__ bool qp_candidate(const QPOptsP& O, double H, double g, double lo, double hi, double x,
                                             double search, double grad, double oldvalue, double step, bool& slow,
                                             double& xc, double& vc)
{
    xc = qp_clamp_fast(lo, hi, x + step * search, slow);
    vc = qp_value1(H, g, xc);
    double sdotg = (xc - x) * grad;
    return qp_quotient_fast(vc - oldvalue, sdotg + 1e-10, O.armijo, slow);
}

// Does the search from x along `search` provably exhaust (see boxqp1_loop)?  Also returns the
// candidate the exhausted loop ends on.  The proof covers every candidate, the first included.
__device__ __forceinline__ bool qp_exhausts(const QPOptsP& O, double H, double g, double lo, double hi, double x,
                                            double search, double grad, bool& slow, double& xl, double& vl)
{
    double xc1 = qp_clamp_fast(lo, hi, x + search, slow); // step 1.0: step * search == search
    double ax = fabs(x), G = fabs(grad), as = fabs(search);
    double gerr = 5e-16 * (fabs(g) + fabs(H * x));
    bool descent = (((search > 0.0) & (grad < 0.0)) | ((search < 0.0) & (grad > 0.0))) & (G > 4.0 * gerr);
    double D0 = 1.001 * as + 4e-16 * ax;
    double Ep = qp_error_budget(O, H, g, ax, G, gerr, D0);
    slow = slow | !(G < __builtin_inf()); // a non-finite gradient defeats "xc == x implies rejected"
    xl = qp_clamp_fast(lo, hi, x + O.step_last * search, slow);
    vl = qp_value1(H, g, xl);
    return (xc1 == x) | qp_spin_proved(O, H, G, gerr, descent, D0, Ep);
}

// mid(): called exactly once, between pass 0 and pass 1 (a caller that shares the work with other wavefronts
// meets them there; see pair_verifier_sweep).
struct QPNoMid {
    __device__ __forceinline__ void operator()() const {}
};
template <class Mid = QPNoMid>
__device__ __forceinline__ void boxqp1(const QPOptsP& O, const double* tab, double H, double sqrtH, double rsqrtH,
                                       double g, double lo, double hi, double x0, QP1Out& out, Mid mid = Mid())
{
    const bool opts_ok = (O.max_iter >= 3) & (O.armijo > 0.0) & (O.armijo < 1.0); // wave-uniform
    if (!opts_ok) {
        mid();
        boxqp1_loop(O, tab, H, sqrtH, rsqrtH, g, lo, hi, x0, out);
        return;
    }
    bool slow = (lo != lo) | (hi != hi) | !(H > 0.0), s;
    const double xa = qp_clamp_fast(lo, hi, x0, slow);
    const double va = qp_value1(H, g, xa);
    const double rhs = g + 0.0;

    // ---- pass 0 (:87-137 with H_ff = ones, then the line search)
    const double grad0 = g + H * xa;
    const bool cl0 = ((xa == lo) & (grad0 > 0.0)) | ((xa == hi) & (grad0 < 0.0));
    const bool e05 = !cl0 & (fabs(grad0) < O.min_grad);
    const bool act0 = !(cl0 | e05);
    const double search0 = rhs * -1 - xa; // (rhs * 1.0) * 1.0 == rhs
    bool jump;
    int tj;
    double sj;
    {   // jump proof (see boxqp1_loop): every trip before tj continues
        double ax = fabs(xa), G = fabs(grad0), as = fabs(search0);
        double gerr = 5e-16 * (fabs(g) + fabs(H * xa));
        bool descent = (((search0 > 0.0) & (grad0 < 0.0)) | ((search0 < 0.0) & (grad0 > 0.0))) & (G > 4.0 * gerr);
        double D0 = 1.001 * as + 4e-16 * ax;
        double Ep = qp_error_budget(O, H, g, ax, G, gerr, D0);
        float Gf = (float)G, Hf = (float)H, cf = (float)(1.0 - O.armijo);
        float cg = cf * Gf;
        float up = (cg + __builtin_amdgcn_sqrtf(cg * cg + (float)(2e-10 * O.armijo) * Hf)) * __builtin_amdgcn_rcpf(Hf);
        float U = fmaxf(up, 1e-10f * __builtin_amdgcn_rcpf(Gf));
        float tf = __builtin_amdgcn_logf(U * __builtin_amdgcn_rcpf((float)as)) * O.inv_log2_dec;
        tf = fminf(fmaxf(tf, 0.0f), 1.0e6f); // NaN -> 0
        tj = (int)tf + 1;
        tj = tj < O.tab_n ? tj : O.tab_n - 1;
        bool try_jump = descent & (tj >= 2);
        tj = try_jump ? tj : 1;
        double sv = tab[tj - 1];
        sj = tab[tj];
        s = false;
        double xv = qp_clamp_fast(lo, hi, xa + sv * search0, s);
        double dv = xv - xa, uv = fabs(dv);
        double hv = (0.5 * H * uv) * uv - ((1.0 - O.armijo) * (G + gerr)) * uv - O.armijo * 1e-10;
        jump = try_jump & (hv > Ep) & (fabs(dv * grad0) > 1.0000001e-10) & !s;
    }
#ifdef ILQR_QP_MID_EARLY
    mid();
#endif
    int n0 = jump ? tj : 0;
    double xb, vb;
    s = false;
    const bool rej0 = qp_candidate(O, H, g, lo, hi, xa, search0, grad0, va, jump ? sj : 1.0, s, xb, vb);
    slow = slow | (act0 & s);
    bool exh0 = false;
    if (__ballot(act0 & rej0) != 0ull) {
        QP_COUNT_WAVE(1);
        double xl, vl;
        s = false;
        bool ex = qp_exhausts(O, H, g, lo, hi, xa, search0, grad0, s, xl, vl);
        exh0 = act0 & rej0 & ex;
        slow = slow | (act0 & rej0 & (s | !ex)); // rejected without proof: more trips
        xb = exh0 ? xl : xb;
        vb = exh0 ? vl : vb;
        n0 = exh0 ? O.n_exhaust : n0;
    }

#ifndef ILQR_QP_MID_EARLY
    mid();
#endif
    // ---- pass 1: factorised Newton step
    const bool act1 = act0 & !exh0;
    const double grad1 = g + H * xb;
    const bool cl1 = ((xb == lo) & (grad1 > 0.0)) | ((xb == hi) & (grad1 < 0.0));
    const bool e16 = act1 & cl1;
    const bool fz = act1 & !cl1; // factorised (:108-117 precede the gradient test)
    const bool e15 = fz & (fabs(grad1) < O.min_grad);
    const bool act1b = fz & !e15;
    const double search1 = ((rhs * rsqrtH) * rsqrtH) * -1 - xb;
    double xc, vc;
    s = false;
    const bool rej1 = qp_candidate(O, H, g, lo, hi, xb, search1, grad1, vb, 1.0, s, xc, vc);
    slow = slow | (act1b & s);
    bool exh1 = false;
    if (__ballot(act1b & rej1) != 0ull) {
        double xl, vl;
        s = false;
        bool ex = qp_exhausts(O, H, g, lo, hi, xb, search1, grad1, s, xl, vl);
        exh1 = act1b & rej1 & ex;
        slow = slow | (act1b & rej1 & (s | !ex));
        xc = exh1 ? xl : xc;
        vc = exh1 ? vl : vc;
    }

    // ---- pass 2: termination tests only (:87-137)
    const bool act2 = act1b & !exh1;
    const double dv2 = vb - vc;
    const bool e24 = act2 & (dv2 < O.min_rel * fabs(vb)) & (dv2 > 0.0);
    const double grad2 = g + H * xc;
    const bool cl2 = ((xc == lo) & (grad2 > 0.0)) | ((xc == hi) & (grad2 < 0.0));
    const bool rest2 = act2 & !e24;
    const bool e26 = rest2 & cl2;
    const bool e25 = rest2 & !cl2 & (fabs(grad2) < O.min_grad);
    slow = slow | (rest2 & !cl2 & !e25); // a third line search
    QP_COUNT(9);
    if (__ballot(slow) != 0ull) {
        QP_COUNT(7);
        QP_COUNT_WAVE(0);
        boxqp1_loop(O, tab, H, sqrtH, rsqrtH, g, lo, hi, x0, out);
        return;
    }
    // outputs, by the stage at which the lane left
    const bool left0 = !act0, left1 = !left0 & (exh0 | e16 | e15); // everything else left in or after pass 1's search
    out.x = left0 ? xa : (left1 ? xb : xc);
    out.rdiag = fz ? rsqrtH : 1.0;
    out.hff = fz ? sqrtH : 1.0;
    out.result = (cl0 | e16 | e26) ? 6 : ((e05 | e15 | e25) ? 5 : ((exh0 | exh1) ? 2 : 4));
    out.free_ = !(cl0 | e16 | e26);
    out.n_iter = (act0 ? 1 : 0) + (act1b ? 1 : 0);
    out.n_factor = fz ? 1 : 0;
    out.n_backtrack = (act0 ? n0 : 0) + (exh1 ? O.n_exhaust : 0);
}

// The outcome of boxqp1 on its most common paths, by the shortest chain of dependent operations: pass 0 jumps
// to trip t_j and accepts, pass 1 leaves at once or accepts the full Newton step, pass 2 ends on a termination test.  Same
// expressions as boxqp1 (a compiler merges the two), none of its checks: the caller must compare with boxqp1.
__device__ __forceinline__ void boxqp1_guess(const QPOptsP& O, const double* tab, double H, double rsqrtH, double g,
                                             double lo, double hi, double x0, double& x_out, bool& free_out)
{
    const double xa = fmax(lo, fmin(hi, x0));
    const double rhs = g + 0.0;
    const double grad0 = g + H * xa;
    const double search0 = rhs * -1 - xa;
    double G = fabs(grad0), as = fabs(search0);
    double gerr = 5e-16 * (fabs(g) + fabs(H * xa));
    bool descent = (((search0 > 0.0) & (grad0 < 0.0)) | ((search0 < 0.0) & (grad0 > 0.0))) & (G > 4.0 * gerr);
    float Gf = (float)G, Hf = (float)H, cf = (float)(1.0 - O.armijo);
    float cg = cf * Gf;
    float up = (cg + __builtin_amdgcn_sqrtf(cg * cg + (float)(2e-10 * O.armijo) * Hf)) * __builtin_amdgcn_rcpf(Hf);
    float U = fmaxf(up, 1e-10f * __builtin_amdgcn_rcpf(Gf));
    float tf = __builtin_amdgcn_logf(U * __builtin_amdgcn_rcpf((float)as)) * O.inv_log2_dec;
    tf = fminf(fmaxf(tf, 0.0f), 1.0e6f);
    int tj = (int)tf + 1;
    tj = tj < O.tab_n ? tj : O.tab_n - 1;
    bool try_jump = descent & (tj >= 2);
    tj = try_jump ? tj : 1;
    const double sj = tab[tj];
    const double xb = fmax(lo, fmin(hi, xa + (try_jump ? sj : 1.0) * search0));
    const double search1 = ((rhs * rsqrtH) * rsqrtH) * -1 - xb;
    const double xc = fmax(lo, fmin(hi, xb + 1.0 * search1));
    const double grad2 = g + H * xc;
    const bool cl2 = ((xc == lo) & (grad2 > 0.0)) | ((xc == hi) & (grad2 < 0.0));
    // (beside the chain: a call that leaves at the start of pass 1 -- on a limit with the gradient pointing outwards,
    // or with a gradient below the threshold -- returns the candidate of pass 0)
    const double grad1 = g + H * xb;
    const bool cl1 = ((xb == lo) & (grad1 > 0.0)) | ((xb == hi) & (grad1 < 0.0));
    const bool left1 = cl1 | (fabs(grad1) < O.min_grad);
    x_out = left1 ? xb : xc;
    free_out = left1 ? !cl1 : !cl2;
}

// ---- backward step ----------------------------------------------------------

struct BwAcc {          // running state of one backward sweep
    double Vx[NX];
    double Vxx[NX * NX];
    double dJ1, dJ2;
    int qp_iter, qp_backtrack, n_clamped;
};

enum { BW_OK = 0, BW_NOT_PD = 1, BW_LINALG = 2 };

template <class Cost>
__device__ __forceinline__ void bw_init(const Cost& C, const double (&xT)[NX], BwAcc& S)
{
    // V_x, V_xx = terminal l_x, l_xx (ilqr_core.py:120-121)
    C.lx_terminal(xT, S.Vx);
#pragma unroll
    for (int a = 0; a < NX; ++a)
#pragma unroll
        for (int b = 0; b < NX; ++b)
            S.Vxx[a * NX + b] = C.lxx_terminal(a, b);
    S.dJ1 = 0.0;
    S.dJ2 = 0.0;
    S.qp_iter = S.qp_backtrack = S.n_clamped = 0;
}

__device__ __forceinline__ int bw_finish(const QPOptsP& QO, const double* tab, const double (&Qx)[NX], double Qu,
                                         const double (&Qxx)[NX * NX], const double (&Qux)[NX], double Quu, double u,
                                         double lower, double upper, BwAcc& S, double& k_out, double (&K_out)[NX],
                                         int& qp_result, bool& qp_free);

// One step i of _backward_pass (ilqr_core.py:127-176) with _compute_Q_terms
// (:179-228) inlined for the cart-pole sparsity pattern and m = 1.
// Returns BW_OK (k_out, K_out, S updated), BW_NOT_PD (the reference prints,
// warns and breaks, :137-140) or BW_LINALG (eigvals would raise on a non-finite
// Q_uu, :291).
template <class Cost>
__device__ __forceinline__ int bw_step(const Cost& C, const QPOptsP& QO, const double* tab, const CartpoleJac& J, double dt,
                                       const double (&x)[NX], double u, double mu, double lower,
                                       double upper, BwAcc& S, double& k_out, double (&K_out)[NX],
                                       int& qp_result, bool& qp_free)
{
    double lx[NX];
    C.lx(x, lx);
    double lu = C.lu(u); // u_diff^T (R + R^T), ilqr_cost.py:586-587

    // column c (>= 2) of f_x is [0, a[0][c-2], a[1][c-2], a[2][c-2], a[3][c-2]]
    const double(&A)[4][3] = J.a;
    const double* Vx = S.Vx;
    const double* Vxx = S.Vxx;

    // Q_x = l_x + f_x^T V_x ; Q_u = l_u + f_u^T V_x      (:206-207)
    double Qx[NX];
    Qx[0] = lx[0] + Vx[0];
    Qx[1] = lx[1] + (dt * Vx[0] + Vx[1]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double acc = A[0][c] * Vx[1];
        acc = fma(A[1][c], Vx[2], acc);
        acc = fma(A[2][c], Vx[3], acc);
        acc = fma(A[3][c], Vx[4], acc);
        Qx[2 + c] = lx[2 + c] + acc;
    }
    double Qu = lu + fma(J.b4, Vx[4], J.b1 * Vx[1]);

    // M1 = f_x^T V_xx ; Q_xx = l_xx + M1 f_x            (:208)
    double M1[NX * NX];
#pragma unroll
    for (int b = 0; b < NX; ++b) {
        M1[0 * NX + b] = Vxx[0 * NX + b];
        M1[1 * NX + b] = dt * Vxx[0 * NX + b] + Vxx[1 * NX + b];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double acc = A[0][c] * Vxx[1 * NX + b];
            acc = fma(A[1][c], Vxx[2 * NX + b], acc);
            acc = fma(A[2][c], Vxx[3 * NX + b], acc);
            acc = fma(A[3][c], Vxx[4 * NX + b], acc);
            M1[(2 + c) * NX + b] = acc;
        }
    }
    double Qxx[NX * NX];
#pragma unroll
    for (int a = 0; a < NX; ++a) {
        Qxx[a * NX + 0] = C.add_lxx(a, 0, M1[a * NX + 0]);
        Qxx[a * NX + 1] = C.add_lxx(a, 1, M1[a * NX + 0] * dt + M1[a * NX + 1]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double acc = M1[a * NX + 1] * A[0][c];
            acc = fma(M1[a * NX + 2], A[1][c], acc);
            acc = fma(M1[a * NX + 3], A[2][c], acc);
            acc = fma(M1[a * NX + 4], A[3][c], acc);
            Qxx[a * NX + 2 + c] = C.add_lxx(a, 2 + c, acc);
        }
    }

    // reg = V_xx + mu I ; M2 = f_u^T reg ; Q_ux = l_ux + M2 f_x ; Q_uu = l_uu + M2 f_u   (:210-212)
    double M2[NX];
#pragma unroll
    for (int b = 0; b < NX; ++b) {
        double r1 = Vxx[1 * NX + b] + (b == 1 ? mu : 0.0);
        double r4 = Vxx[4 * NX + b] + (b == 4 ? mu : 0.0);
        M2[b] = fma(J.b4, r4, J.b1 * r1);
    }
    double Qux[NX];
    Qux[0] = 0.0 + M2[0];
    Qux[1] = 0.0 + (M2[0] * dt + M2[1]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double acc = M2[1] * A[0][c];
        acc = fma(M2[2], A[1][c], acc);
        acc = fma(M2[3], A[2][c], acc);
        acc = fma(M2[4], A[3][c], acc);
        Qux[2 + c] = 0.0 + acc;
    }
    double Quu = C.luu() + fma(M2[4], J.b4, M2[1] * J.b1);
    return bw_finish(QO, tab, Qx, Qu, Qxx, Qux, Quu, u, lower, upper, S, k_out, K_out, qp_result, qp_free);
}

// The part of a backward step that follows the Q-terms: PD test, boxQP, gains, value update.
__device__ __forceinline__ int bw_finish(const QPOptsP& QO, const double* tab, const double (&Qx)[NX], double Qu,
                                         const double (&Qxx)[NX * NX], const double (&Qux)[NX], double Quu, double u,
                                         double lower, double upper, BwAcc& S, double& k_out, double (&K_out)[NX],
                                         int& qp_result, bool& qp_free)
{
    // np.all(np.linalg.eigvals(Q_uu) > 0), then cholesky   (:137-141)
    if (!(fabs(Quu) < __builtin_inf()))
        return BW_LINALG;
    if (!(Quu > 0.0))
        return BW_NOT_PD;
    double L = sqrt(Quu);
    double rL = 1.0 / L;
    // k[i] = cho_solve((R_Q_uu, True), Q_u): the (positive-sign) boxQP warm start   (:145)
    double k0 = (Qu * rL) * rL;

    // boxQP(Q_uu, Q_u, lower - u, upper - u, k[i], ...)     (:152-156)
    QP1Out qp;
#ifdef ILQR_ABLATE_QP // timing diagnostic only (wrong numerics): the clamped Newton point instead of boxQP
    qp.x = clamp1(lower - u, upper - u, k0 * -1);
    qp.rdiag = rL;
    qp.hff = L;
    qp.result = 5;
    qp.free_ = (qp.x > lower - u) & (qp.x < upper - u);
    qp.n_iter = qp.n_factor = qp.n_backtrack = 0;
#else
    boxqp1(QO, tab, Quu, L, rL, Qu, lower - u, upper - u, k0, qp);
#endif
    double kk = qp.x;
    S.qp_iter += qp.n_iter;
    S.qp_backtrack += qp.n_backtrack;
    S.n_clamped += qp.free_ ? 0 : 1;
    qp_result = qp.result;
    qp_free = qp.free_;

    // K[i, free, :] = -cho_solve((H_ff, True), Q_ux[free, :]); clamped rows stay 0   (:157-160)
    double K[NX];
#pragma unroll
    for (int b = 0; b < NX; ++b)
        K[b] = qp.free_ ? ((Qux[b] * qp.rdiag) * qp.rdiag) * -1 : 0.0;

    // V_x = Q_x + K^T Q_uu k + K^T Q_u + Q_ux^T k                    (:164)
    // V_xx = Q_xx + K^T Q_uu K + K^T Q_ux + Q_ux^T K, symmetrised    (:167-169)
    double T1[NX];
#pragma unroll
    for (int a = 0; a < NX; ++a)
        T1[a] = K[a] * Quu;
    double Vn[NX * NX];
#pragma unroll
    for (int a = 0; a < NX; ++a) {
        S.Vx[a] = ((Qx[a] + T1[a] * kk) + K[a] * Qu) + Qux[a] * kk;
#pragma unroll
        for (int b = 0; b < NX; ++b)
            Vn[a * NX + b] = ((Qxx[a * NX + b] + T1[a] * K[b]) + K[a] * Qux[b]) + Qux[a] * K[b];
    }
    // 0.5 * (V + V^T): entries (a, b) and (b, a) are the same sum, computed once
#pragma unroll
    for (int a = 0; a < NX; ++a)
#pragma unroll
        for (int b = a; b < NX; ++b) {
            double v = 0.5 * (Vn[a * NX + b] + Vn[b * NX + a]);
            S.Vxx[a * NX + b] = v;
            S.Vxx[b * NX + a] = v;
        }

    // Delta_J1 += k.Q_u ; Delta_J2 += 0.5 k.(Q_uu k)                 (:175-176)
    S.dJ1 = S.dJ1 + kk * Qu;
    S.dJ2 = S.dJ2 + (kk * (Quu * kk)) * 0.5;

    k_out = kk;
#pragma unroll
    for (int b = 0; b < NX; ++b)
        K_out[b] = K[b];
    return BW_OK;
}

// The same step from derivative tensors handed in as dense arrays (the streams a derivative
// kernel writes): _compute_Q_terms (ilqr_core.py:179-228) with every term of every np.dot,
// f_x [5][5], f_u [5], l_x [5], l_u, l_xx [5][5], l_uu, l_ux [5].
__device__ __forceinline__ int bw_step_dense(const QPOptsP& QO, const double* tab, const double (&fx)[NX * NX],
                                             const double (&fu)[NX], const double (&lx)[NX], double lu,
                                             const double (&lxx)[NX * NX], double luu, const double (&lux)[NX],
                                             double u, double mu, double lower, double upper, BwAcc& S,
                                             double& k_out, double (&K_out)[NX], int& qp_result, bool& qp_free)
{
    const double* Vx = S.Vx;
    const double* Vxx = S.Vxx;
    double Qx[NX], Qxx[NX * NX], Qux[NX], M1[NX * NX], M2[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        double acc = fx[j] * Vx[0];
#pragma unroll
        for (int a = 1; a < NX; ++a)
            acc = fma(fx[a * NX + j], Vx[a], acc);
        Qx[j] = lx[j] + acc;
    }
    double Qu;
    {
        double acc = fu[0] * Vx[0];
#pragma unroll
        for (int a = 1; a < NX; ++a)
            acc = fma(fu[a], Vx[a], acc);
        Qu = lu + acc;
    }
#pragma unroll
    for (int a = 0; a < NX; ++a)
#pragma unroll
        for (int b = 0; b < NX; ++b) {
            double acc = fx[a] * Vxx[b];
#pragma unroll
            for (int c = 1; c < NX; ++c)
                acc = fma(fx[c * NX + a], Vxx[c * NX + b], acc);
            M1[a * NX + b] = acc;
        }
#pragma unroll
    for (int a = 0; a < NX; ++a)
#pragma unroll
        for (int b = 0; b < NX; ++b) {
            double acc = M1[a * NX] * fx[b];
#pragma unroll
            for (int c = 1; c < NX; ++c)
                acc = fma(M1[a * NX + c], fx[c * NX + b], acc);
            Qxx[a * NX + b] = lxx[a * NX + b] + acc;
        }
#pragma unroll
    for (int b = 0; b < NX; ++b) {
        double acc = fu[0] * (Vxx[b] + mu * (b == 0 ? 1.0 : 0.0));
#pragma unroll
        for (int c = 1; c < NX; ++c)
            acc = fma(fu[c], Vxx[c * NX + b] + mu * (b == c ? 1.0 : 0.0), acc);
        M2[b] = acc;
    }
#pragma unroll
    for (int b = 0; b < NX; ++b) {
        double acc = M2[0] * fx[b];
#pragma unroll
        for (int c = 1; c < NX; ++c)
            acc = fma(M2[c], fx[c * NX + b], acc);
        Qux[b] = lux[b] + acc;
    }
    double Quu;
    {
        double acc = M2[0] * fu[0];
#pragma unroll
        for (int c = 1; c < NX; ++c)
            acc = fma(M2[c], fu[c], acc);
        Quu = luu + acc;
    }
    return bw_finish(QO, tab, Qx, Qu, Qxx, Qux, Quu, u, lower, upper, S, k_out, K_out, qp_result, qp_free);
}

// u_new = u + alpha k + K (x_new - x), ilqr_core.py:104 (never clamped)
__device__ __forceinline__ double fw_control(const double (&xn)[NX], const double (&xo)[NX], double u,
                                             double kk, const double (&K)[NX], double alpha)
{
    double acc = K[0] * (xn[0] - xo[0]);
#pragma unroll
    for (int a = 1; a < NX; ++a)
        acc = fma(K[a], xn[a] - xo[a], acc);
    return (u + alpha * kk) + acc;
}

} // namespace ilqr
