// User-defined dynamics for the batched control-limited iLQR solve (up to 8 states; one control
// on the fast path, 2..4 controls on a general path, see ILQR_USER_M below).  The reference gets its generality from Python callbacks (AutogradDynamics /
// FiniteDifferenceDynamics, ilqr/ilqr_dynamics.py:94-575), which cannot run inside a kernel;
// here a model is a small C source with two functions,
//
//     ILQR_FN void ilqr_user_f  (const double* p, const double* x, double u, double* out);
//     ILQR_FN void ilqr_user_jac(const double* p, const double* x, double u, double* fx, double* fu);
//
// (f, and f_x [N][N] row-major / f_u [N]) plus  #define ILQR_USER_N  and  ILQR_USER_NP  (number of
// parameters p), and this file is compiled ONCE PER MODEL around it:
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
//           -DILQR_USER_MODEL='"model.h"' ilqr_user_tu.hip -o libilqr_user_<name>.so
//
// The model may call ilqr_sincos(x, &s, &c), ilqr_atan2(y, x), ilqr_sqrt(x): the library's own
// deterministic functions.  The solver is iLQR.update_u (ilqr_core.py:241-399) with the dense
// _compute_Q_terms (every term of every np.dot), boxQP and the serial first-hit line search,
// one problem per lane, arrays batch-minor in a workspace.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../include/ilqr_hip.h"
#include "../../include/ilqr_user.h"
#include "ilqr_device.hpp"
#include "ilqr_qp_dense.hpp"

#define ILQR_FN __device__ __forceinline__
#define ilqr_sincos(x, s, c) ilqr::det_sincos((x), *(s), *(c))
#define ilqr_atan2(y, x) ilqr::det_atan2((y), (x))
#define ilqr_sqrt(x) sqrt(x)
#ifndef ILQR_USER_MODEL
#error "compile with -DILQR_USER_MODEL='\"path/to/model.h\"'"
#endif
#include ILQR_USER_MODEL
#ifndef ILQR_USER_M
#define ILQR_USER_M 1 // controls; with more than one, u is passed to the model as a pointer
#endif
#include "../../include/ilqr_user_fd.h" // ilqr_user_jac by forward differences if the model asks for it

namespace {

using namespace ilqr;

constexpr int UN = ILQR_USER_N;
constexpr int UM = ILQR_USER_M;
constexpr int UWG = 4; // most wavefronts per workgroup of the solve kernels
constexpr int UNP = ILQR_USER_NP > 0 ? ILQR_USER_NP : 1;
static_assert(UN >= 1 && UN <= 8, "ILQR_USER_N must be between 1 and 8");
static_assert(UM >= 1 && UM <= 4, "ILQR_USER_M must be between 1 and 4");

struct UParams {
    double p[UNP];
};

// rows of 64 doubles in one wavefront's workspace tile: two copies of xs, us, k, K
__host__ __device__ inline int64_t utile_rows(int64_t T) { return 2 * ((T + 1) * UN + T * UM + T * UM + T * UM * UN); }

#if ILQR_USER_M == 1
struct UCost { // QRCost (ilqr/ilqr_cost.py:490-530) or PathQRCost (:635-770), m = 1
    double Q[UN * UN], Qt[UN * UN], QQ[UN * UN], QQt[UN * UN];
    double R, RR, xg[UN], ug;
    const double *x_path, *u_path; // device goals, or null (the goal is xg, ug): PathQRCost [T+1][n], [T] (goals by
                                   // time step, value without 1/2) and / or one goal or path per problem
    int path;                      // 1: the goals depend on the time step (PathQRCost)
    int64_t gsx, gsu;              // doubles between the goals of consecutive problems (0: shared by the batch)
};
// goal of problem gp at time step t (wave-uniform loads when the batch shares one path)
__device__ __forceinline__ void ugoal_x(const UCost& C, int64_t gp, int t, double (&xg)[UN])
{
    const double* g = C.x_path != nullptr ? C.x_path + gp * C.gsx + (C.path ? (int64_t)t * UN : 0) : C.xg;
#pragma unroll
    for (int i = 0; i < UN; ++i)
        xg[i] = g[i];
}
__device__ __forceinline__ double ugoal_u(const UCost& C, int64_t gp, int t)
{
    return C.u_path != nullptr ? C.u_path[gp * C.gsu + (C.path ? t : 0)] : C.ug;
}
struct UBounds {
    double lo, hi;
};

__device__ __forceinline__ void uload(const double* base, int64_t ld, int64_t row0, int64_t b, double (&v)[UN])
{
#pragma unroll
    for (int i = 0; i < UN; ++i)
        v[i] = base[(row0 + i) * ld + b];
}
__device__ __forceinline__ void ustore(double* base, int64_t ld, int64_t row0, int64_t b, const double (&v)[UN])
{
#pragma unroll
    for (int i = 0; i < UN; ++i)
        base[(row0 + i) * ld + b] = v[i];
}

// (v^T M) . v in the order of x_diff.T.dot(Q).dot(x_diff), ilqr_cost.py:547
__device__ __forceinline__ double uquad(const double (&v)[UN], const double* M)
{
    double tot = 0.0;
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        double acc = v[0] * M[j];
#pragma unroll
        for (int i = 1; i < UN; ++i)
            acc = fma(v[i], M[i * UN + j], acc);
        tot = (j == 0) ? acc * v[0] : fma(acc, v[j], tot);
    }
    return tot;
}
__device__ __forceinline__ double ucost_l(const UCost& C, const double (&x)[UN], double u, int64_t gp, int t)
{
    double xd[UN], xg[UN];
    ugoal_x(C, gp, t, xg);
#pragma unroll
    for (int i = 0; i < UN; ++i)
        xd[i] = x[i] - xg[i];
    double sx = uquad(xd, C.Q);
    double ud = u - ugoal_u(C, gp, t);
    double su = (ud * C.R) * ud;
    if (!C.path) { // QRCost.l carries the 1/2 (ilqr_cost.py:547-552), PathQRCost.l does not (:695-701)
        sx = sx * 0.5;
        su = su * 0.5;
    }
    return sx + su;
}
__device__ __forceinline__ double ucost_lT(const UCost& C, const double (&x)[UN], int64_t gp, int T)
{
    double xd[UN], xg[UN];
    ugoal_x(C, gp, T, xg);
#pragma unroll
    for (int i = 0; i < UN; ++i)
        xd[i] = x[i] - xg[i];
    double sx = uquad(xd, C.Qt);
    return C.path ? sx : sx * 0.5;
}
__device__ __forceinline__ void ucost_lx(const double* QQ, const UCost& C, const double (&x)[UN], int64_t gp, int t,
                                         double (&out)[UN])
{
    double xd[UN], xg[UN];
    ugoal_x(C, gp, t, xg);
#pragma unroll
    for (int i = 0; i < UN; ++i)
        xd[i] = x[i] - xg[i];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        double acc = xd[0] * QQ[j];
#pragma unroll
        for (int a = 1; a < UN; ++a)
            acc = fma(xd[a], QQ[a * UN + j], acc);
        out[j] = acc;
    }
}

struct UAcc {
    double Vx[UN], Vxx[UN * UN];
    double dJ1, dJ2;
    int qp_iter, qp_backtrack, n_clamped;
};

// One step of _backward_pass (ilqr_core.py:127-176) with the dense _compute_Q_terms (:179-228).
__device__ __forceinline__ int ubw_step(const UParams& P, const UCost& C, const QPOptsP& QO, const double* tab,
                                        const double (&x)[UN], double u, int64_t gp, int t, double mu, double lower,
                                        double upper, UAcc& S, double& k_out, double (&K_out)[UN])
{
    double fx[UN * UN], fu[UN], lx[UN];
    ilqr_user_jac(P.p, x, u, fx, fu);
    ucost_lx(C.QQ, C, x, gp, t, lx);
    double lu = (u - ugoal_u(C, gp, t)) * C.RR;
    const double* Vx = S.Vx;
    const double* Vxx = S.Vxx;
    double Qx[UN], Qxx[UN * UN], Qux[UN], M1[UN * UN], M2[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        double acc = fx[j] * Vx[0];
#pragma unroll
        for (int a = 1; a < UN; ++a)
            acc = fma(fx[a * UN + j], Vx[a], acc);
        Qx[j] = lx[j] + acc;
    }
    double Qu;
    {
        double acc = fu[0] * Vx[0];
#pragma unroll
        for (int a = 1; a < UN; ++a)
            acc = fma(fu[a], Vx[a], acc);
        Qu = lu + acc;
    }
#pragma unroll
    for (int a = 0; a < UN; ++a)
#pragma unroll
        for (int b = 0; b < UN; ++b) {
            double acc = fx[a] * Vxx[b];
#pragma unroll
            for (int c = 1; c < UN; ++c)
                acc = fma(fx[c * UN + a], Vxx[c * UN + b], acc);
            M1[a * UN + b] = acc;
        }
#pragma unroll
    for (int a = 0; a < UN; ++a)
#pragma unroll
        for (int b = 0; b < UN; ++b) {
            double acc = M1[a * UN] * fx[b];
#pragma unroll
            for (int c = 1; c < UN; ++c)
                acc = fma(M1[a * UN + c], fx[c * UN + b], acc);
            Qxx[a * UN + b] = C.QQ[a * UN + b] + acc;
        }
#pragma unroll
    for (int b = 0; b < UN; ++b) {
        double acc = fu[0] * (Vxx[b] + mu * (b == 0 ? 1.0 : 0.0));
#pragma unroll
        for (int c = 1; c < UN; ++c)
            acc = fma(fu[c], Vxx[c * UN + b] + mu * (b == c ? 1.0 : 0.0), acc);
        M2[b] = acc;
    }
#pragma unroll
    for (int b = 0; b < UN; ++b) {
        double acc = M2[0] * fx[b];
#pragma unroll
        for (int c = 1; c < UN; ++c)
            acc = fma(M2[c], fx[c * UN + b], acc);
        Qux[b] = 0.0 + acc;
    }
    double Quu;
    {
        double acc = M2[0] * fu[0];
#pragma unroll
        for (int c = 1; c < UN; ++c)
            acc = fma(M2[c], fu[c], acc);
        Quu = C.RR + acc;
    }
    // np.all(np.linalg.eigvals(Q_uu) > 0), then cholesky   (:137-141)
    if (!(fabs(Quu) < __builtin_inf()))
        return BW_LINALG;
    if (!(Quu > 0.0))
        return BW_NOT_PD;
    double L = sqrt(Quu);
    double rL = 1.0 / L;
    double k0 = (Qu * rL) * rL; // the (positive-sign) boxQP warm start (:145)
    QP1Out qp;
    boxqp1(QO, tab, Quu, L, rL, Qu, lower - u, upper - u, k0, qp);
    double kk = qp.x;
    S.qp_iter += qp.n_iter;
    S.qp_backtrack += qp.n_backtrack;
    S.n_clamped += qp.free_ ? 0 : 1;
    double K[UN], T1[UN], Vn[UN * UN];
#pragma unroll
    for (int b = 0; b < UN; ++b) {
        K[b] = qp.free_ ? ((Qux[b] * qp.rdiag) * qp.rdiag) * -1 : 0.0;
        T1[b] = K[b] * Quu;
    }
#pragma unroll
    for (int a = 0; a < UN; ++a) {
        S.Vx[a] = ((Qx[a] + T1[a] * kk) + K[a] * Qu) + Qux[a] * kk;
#pragma unroll
        for (int b = 0; b < UN; ++b)
            Vn[a * UN + b] = ((Qxx[a * UN + b] + T1[a] * K[b]) + K[a] * Qux[b]) + Qux[a] * K[b];
    }
#pragma unroll
    for (int a = 0; a < UN; ++a)
#pragma unroll
        for (int b = a; b < UN; ++b) {
            double v = 0.5 * (Vn[a * UN + b] + Vn[b * UN + a]);
            S.Vxx[a * UN + b] = v;
            S.Vxx[b * UN + a] = v;
        }
    S.dJ1 = S.dJ1 + kk * Qu;
    S.dJ2 = S.dJ2 + (kk * (Quu * kk)) * 0.5;
    k_out = kk;
#pragma unroll
    for (int b = 0; b < UN; ++b)
        K_out[b] = K[b];
    return BW_OK;
}

// One backward sweep; gains of the steps it does not reach are zero (np.zeros_like, :118-119).
__device__ int ubackward(const UParams& P, const UCost& C, const QPOptsP& QO, const double* tab, const UBounds& bd,
                         int T, int64_t ld, int64_t b, int64_t gp0, const double* xs, const double* us, double mu,
                         double* k, double* K, UAcc& S, int& stopped_at)
{
    const int64_t gp = gp0 + b; // the problem in column b of the tile: its goal may be its own
    double xT[UN];
    uload(xs, ld, (int64_t)T * UN, b, xT);
    ucost_lx(C.QQt, C, xT, gp, T, S.Vx);
#pragma unroll
    for (int i = 0; i < UN * UN; ++i)
        S.Vxx[i] = C.QQt[i];
    S.dJ1 = 0.0;
    S.dJ2 = 0.0;
    S.qp_iter = S.qp_backtrack = S.n_clamped = 0;
    stopped_at = -1;
    int rc = BW_OK;
    for (int i = T - 1; i >= 0; --i) {
        double kk = 0.0, Kr[UN];
#pragma unroll
        for (int a = 0; a < UN; ++a)
            Kr[a] = 0.0;
        if (rc == BW_OK) {
            double x[UN], kt, Kt[UN];
            uload(xs, ld, (int64_t)i * UN, b, x);
            double u = us[(int64_t)i * ld + b];
            int r = ubw_step(P, C, QO, tab, x, u, gp, i, mu, bd.lo, bd.hi, S, kt, Kt);
            if (r == BW_OK) {
                kk = kt;
#pragma unroll
                for (int a = 0; a < UN; ++a)
                    Kr[a] = Kt[a];
            } else {
                rc = r;
                if (r == BW_NOT_PD)
                    stopped_at = i;
            }
        }
        k[(int64_t)i * ld + b] = kk;
        ustore(K, ld, (int64_t)i * UN, b, Kr);
    }
    return rc;
}

// _forward_pass + _trajectory_cost (ilqr_core.py:85-107, :230-239); k == nullptr: plain rollout of us.
__device__ double uforward(const UParams& P, const UCost& C, int T, int64_t ld, int64_t b, int64_t gp0,
                           const double* xs, const double* us, const double* k, const double* K, double alpha,
                           const double* x0, double* xs_new, double* us_new)
{
    const int64_t gp = gp0 + b;
    double xn[UN], xt[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i)
        xn[i] = x0[(int64_t)i * ld + b];
    double J = 0.0;
    for (int t = 0; t < T; ++t) {
        double u = us[(int64_t)t * ld + b];
        double un = u;
        if (k != nullptr) {
            double xo[UN], Kr[UN];
            uload(xs, ld, (int64_t)t * UN, b, xo);
            uload(K, ld, (int64_t)t * UN, b, Kr);
            double acc = Kr[0] * (xn[0] - xo[0]);
#pragma unroll
            for (int a = 1; a < UN; ++a)
                acc = fma(Kr[a], xn[a] - xo[a], acc);
            un = (u + alpha * k[(int64_t)t * ld + b]) + acc; // never clamped (:104)
        }
        if (xs_new != nullptr) {
            ustore(xs_new, ld, (int64_t)t * UN, b, xn);
            us_new[(int64_t)t * ld + b] = un;
        }
        J = J + ucost_l(C, xn, un, gp, t);
        ilqr_user_f(P.p, xn, un, xt);
#pragma unroll
        for (int a = 0; a < UN; ++a)
            xn[a] = xt[a];
    }
    if (xs_new != nullptr)
        ustore(xs_new, ld, (int64_t)T * UN, b, xn);
    return J + ucost_lT(C, xn, gp, T);
}

// Workspace of one wavefront's 64 problems: rows of 64 doubles (lane = problem), two copies of
// everything that is swapped, all in one contiguous tile so that a wave walks its own region.
struct UWork {
    double *xs0, *xs1, *us0, *us1, *k0, *k1, *K0, *K1;
};

__device__ __forceinline__ UWork utile(double* ws, int64_t tile, int64_t T)
{
    double* p = ws + tile * utile_rows(T) * 64;
    UWork W;
    W.xs0 = p;
    W.xs1 = W.xs0 + (T + 1) * UN * 64;
    W.us0 = W.xs1 + (T + 1) * UN * 64;
    W.us1 = W.us0 + T * 64;
    W.k0 = W.us1 + T * 64;
    W.k1 = W.k0 + T * 64;
    W.K0 = W.k1 + T * 64;
    W.K1 = W.K0 + T * UN * 64;
    return W;
}

constexpr int ULS_PASS = 8;  // problems per cooperative line-search pass
constexpr int ULS_MAXA = 32; // largest alpha schedule

// iLQR.update_u (ilqr_core.py:241-399), every lane its own problem.  Reference-layout arrays are
// copied into / out of the wave's workspace tile by the lane itself.
// Line search (:301-313): every lane rolls out alpha_0 for its own problem; the problems that
// need more hand their remaining candidates to the whole wave (lane = one (problem, alpha) pair,
// costs into LDS) and then scan the costs in the reference's order.  A rollout is a pure function
// of (problem, alpha), so decisions and numbers are those of the serial loop; a rejected
// iteration costs ~3 rollout times instead of 30.
__global__ __launch_bounds__(256, 1) void k_user_solve(UParams P, UCost C_arg, SolveOptsP O, UBounds bd, int T, int64_t B,
                                                   double* ws, const double* x0, const double* us_init,
                                                   double* delta_io, double* k_io, double* K_io, double* xs_out,
                                                   double* us_out, double* J_out, double* mu_out,
                                                   int32_t* iters_out, int32_t* status_out, ilqr_trace tr)
{
    __shared__ double s_qp_tab[QP_TAB];
    __shared__ double s_alpha[ULS_MAXA];
    // four wavefronts per workgroup, each with its own slice of the line-search arrays (a wave only talks
    // to itself): packed four to a CU the waves of a small batch run more evenly than one per CU
    __shared__ double s_J_all[UWG][ULS_PASS][ULS_MAXA];
    __shared__ int s_lane_all[UWG][ULS_PASS], s_cur_all[UWG][ULS_PASS], s_gcur_all[UWG][ULS_PASS];
    // The cost matrices are read from LDS (wave-uniform addresses): held in scalar registers for the whole
    // kernel, their ~220 registers were spilled to vector-register lanes and a quarter of all vector
    // instructions were the lane reads that fetched them back.
    __shared__ UCost s_cost;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double(&s_J)[ULS_PASS][ULS_MAXA] = s_J_all[wv];
    int(&s_lane)[ULS_PASS] = s_lane_all[wv];
    int(&s_cur)[ULS_PASS] = s_cur_all[wv];
    int(&s_gcur)[ULS_PASS] = s_gcur_all[wv];
    if (threadIdx.x == 0)
        s_cost = C_arg;
    const UCost& C = s_cost;
    if (threadIdx.x < ULS_MAXA)
        s_alpha[threadIdx.x] = O.alphas[threadIdx.x];
    qp_fill_step_table(s_qp_tab, O.qp.step_dec); // ends with the workgroup barrier
    // problems per wavefront: the first O.lanes_per_wave columns of the wave's 64-column tile are in use
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const int64_t gp0 = tile * O.lanes_per_wave; // problem in column 0 of the tile
    const int64_t b = gp0 + lane;
    const bool valid = lane < O.lanes_per_wave && b < B;
    const UWork W = utile(ws, tile, T);
    const int64_t ld = 64, bl = lane; // row stride and column inside the tile
#define XS(i) ((i) ? W.xs1 : W.xs0)
#define US(i) ((i) ? W.us1 : W.us0)
#define KF(i) ((i) ? W.k1 : W.k0)
#define KG(i) ((i) ? W.K1 : W.K0)
    QPOptsP QO = qp_opts_in_vgprs(O.qp);
    int cur = 0, gcur = 0;
    double mu = 1.0, delta = 2.0, J = 0.0;
    if (valid) {
        for (int i = 0; i < UN; ++i)
            W.xs0[(int64_t)i * ld + bl] = x0[b * UN + i];
        for (int t = 0; t < T; ++t) {
            W.us0[(int64_t)t * ld + bl] = us_init[b * T + t];
            W.k0[(int64_t)t * ld + bl] = k_io[b * T + t];
            for (int i = 0; i < UN; ++i)
                W.K0[((int64_t)t * UN + i) * ld + bl] = K_io[(b * T + t) * UN + i];
        }
        mu = O.mu_from_array ? mu_out[b] : 1.0; // reset per call (:264); delta_ persists (:62)
        delta = delta_io[b];
        // initial rollout and cost (:269-271)
        J = uforward(P, C, T, ld, bl, gp0, nullptr, W.us0, nullptr, nullptr, 0.0, W.xs0, W.xs0, W.us0);
    }
    double dJ1 = 0.0, dJ2 = 0.0, actual = 0.0;
    if (valid && O.mu_from_array && O.dJ_carry != nullptr) { // a continued solve: the sums of its last backward pass
        dJ1 = O.dJ_carry[b];
        dJ2 = O.dJ_carry[B + b];
    }
    int status = ILQR_RAN_OUT, iters = 0;
    bool running = valid && O.max_iter > 0;
    const int n_more = O.n_alpha - 1; // candidates after alpha_0
    for (int it = 0; it < O.max_iter; ++it) {
        if (__ballot(running) == 0ull)
            break;
        UAcc S;
        int stopped_at = -1;
        S.qp_iter = S.qp_backtrack = S.n_clamped = 0;
        int accepted = 0, j = 0;
        double Jnew = 0.0;
        const int nxt = cur ^ 1;
        bool need = false; // the serial loop would go on to alpha_1
        if (running) {
            // backward pass with the LinAlgError retry loop (:287-299)
            for (;;) {
                int gnew = gcur ^ 1;
                int rc = ubackward(P, C, QO, s_qp_tab, bd, T, ld, bl, gp0, XS(cur), US(cur), mu, KF(gnew), KG(gnew), S,
                                   stopped_at);
                if (rc != BW_LINALG) {
                    gcur = gnew;
                    dJ1 = S.dJ1;
                    dJ2 = S.dJ2;
                    break;
                }
                delta = fmax(delta, 1.0) * O.delta0;
                mu = fmax(O.mu_min, mu * delta);
                if (O.mu_max != 0.0 && mu >= O.mu_max)
                    break;
            }
            // line search, alpha_0 (:301-313)
            if (O.n_alpha > 0) {
                double alpha = s_alpha[0];
                Jnew = uforward(P, C, T, ld, bl, gp0, XS(cur), US(cur), KF(gcur), KG(gcur), alpha, XS(cur), XS(nxt), US(nxt));
                double expected = (alpha * dJ1 + alpha * alpha * dJ2) * -1;
                if (expected <= 0.0) {
                    // reject the whole iteration
                } else if ((J - Jnew) / expected > O.scale) {
                    actual = J - Jnew;
                    accepted = 1;
                } else {
                    need = n_more > 0;
                    j = need ? 1 : O.n_alpha;
                }
            }
        }
        // remaining alphas of the problems that need them, ULS_PASS problems per pass
        // (helper lanes read gains another lane of this wave has just stored)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        unsigned long long mask = __ballot(need);
        bool redo = false; // accepted at some alpha_j, j >= 1: its trajectory still has to be written
        while (mask != 0ull) {
            int rank = __popcll(mask & ((1ull << lane) - 1ull));
            bool mine = need && rank < ULS_PASS;
            if (mine) {
                s_lane[rank] = lane;
                s_cur[rank] = cur;
                s_gcur[rank] = gcur;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int np = __popcll(mask);
            np = np < ULS_PASS ? np : ULS_PASS;
            int nc = np * n_more;
            for (int q0 = 0; q0 < nc; q0 += 64) {
                int q = q0 + lane;
                if (q < nc) {
                    int fi = q / n_more, jj = 1 + q % n_more;
                    int64_t pb = s_lane[fi];
                    int pc = s_cur[fi], pg = s_gcur[fi];
                    s_J[fi][jj] = uforward(P, C, T, ld, pb, gp0, XS(pc), US(pc), KF(pg), KG(pg), s_alpha[jj], XS(pc), nullptr,
                                           nullptr);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (mine) {
                // the reference's loop from alpha_1 on, on the tabulated costs
                for (j = 1; j < O.n_alpha; ++j) {
                    double alpha = s_alpha[j];
                    Jnew = s_J[rank][j];
                    double expected = (alpha * dJ1 + alpha * alpha * dJ2) * -1;
                    if (expected <= 0.0) {
                        break;
                    } else if ((J - Jnew) / expected > O.scale) {
                        actual = J - Jnew;
                        accepted = 1;
                        redo = true;
                        break;
                    }
                }
                need = false;
            }
            __builtin_amdgcn_wave_barrier();
            mask = __ballot(need);
        }
        if (redo) // the accepted candidate's states and controls (same arithmetic, same bits)
            Jnew = uforward(P, C, T, ld, bl, gp0, XS(cur), US(cur), KF(gcur), KG(gcur), s_alpha[j], XS(cur), XS(nxt), US(nxt));
        if (running) {
            bool stop = false;
            int converged = 0;
            if (accepted) {
                delta = fmin(1.0, delta) / O.delta0;
                mu = mu * delta;
                if (mu <= O.mu_min)
                    mu = 0.0;
                J = Jnew;
                cur = nxt;
                if (actual < O.tol)
                    converged = 1;
            } else {
                delta = fmax(delta, 1.0) * O.delta0;
                mu = fmax(O.mu_min, mu * delta);
                if (O.mu_max != 0.0 && mu >= O.mu_max) {
                    status = ILQR_MU_MAX;
                    stop = true;
                }
            }
            int64_t to = (int64_t)it * B + b;
            if (tr.J) tr.J[to] = J;
            if (tr.accepted) tr.accepted[to] = accepted;
            if (tr.alpha_idx) tr.alpha_idx[to] = j;
            if (tr.mu) tr.mu[to] = mu;
            if (tr.delta) tr.delta[to] = delta;
            if (tr.qp_iter) tr.qp_iter[to] = S.qp_iter;
            if (tr.qp_backtrack) tr.qp_backtrack[to] = S.qp_backtrack;
            if (tr.n_clamped) tr.n_clamped[to] = S.n_clamped;
            if (tr.stopped_at) tr.stopped_at[to] = stopped_at;
            iters = it + 1;
            if (converged)
                status = ILQR_CONVERGED;
            if (stop || converged)
                running = false;
        }
    }
    if (valid) {
        const double* xf = XS(cur);
        const double* uf = US(cur);
        const double* kf = KF(gcur);
        const double* Kf = KG(gcur);
        for (int t = 0; t <= T; ++t)
            for (int i = 0; i < UN; ++i)
                xs_out[(b * (T + 1) + t) * UN + i] = xf[((int64_t)t * UN + i) * ld + bl];
        for (int t = 0; t < T; ++t) {
            us_out[b * T + t] = uf[(int64_t)t * ld + bl];
            k_io[b * T + t] = kf[(int64_t)t * ld + bl];
            for (int i = 0; i < UN; ++i)
                K_io[(b * T + t) * UN + i] = Kf[((int64_t)t * UN + i) * ld + bl];
        }
        delta_io[b] = delta;
        J_out[b] = J;
        mu_out[b] = mu;
        iters_out[b] = iters;
        status_out[b] = status;
        if (O.mu_from_array && O.dJ_carry != nullptr) { // only a solve that is being continued carries the sums
            O.dJ_carry[b] = dJ1;
            O.dJ_carry[B + b] = dJ2;
        }
    }
#undef XS
#undef US
#undef KF
#undef KG
}

#else // ---------------------------------------------------------------- more than one control

// The general case, 2 <= m <= 4: a direct per-lane evaluation of _backward_pass / _compute_Q_terms /
// _forward_pass (ilqr_core.py:85-228) with every term of every np.dot as a left-to-right chain, the
// dense boxQP (boxqp_dense), and the cooperative line search of the single-control kernel.  Written for
// generality first: the per-problem matrices of the larger sizes live partly in scratch memory.

struct GCost { // QRCost (ilqr/ilqr_cost.py:490-530) or PathQRCost (:635-770)
    double Q[UN * UN], Qt[UN * UN], R[UM * UM], xg[UN], ug[UM];
    const double *x_path, *u_path; // device goals or null (see UCost): [T+1][n], [T][m] per path
    int path;
    int64_t gsx, gsu;
};
// The goal of problem gp at time step t, into registers.  (Two separate loads, not one load through a selected
// pointer: the shared goal lives in LDS with the cost, a path or a goal per problem in global memory, and a pointer
// that may be either is a generic pointer -- its loads are slower and every one of them was waited for at once.)
typedef const double __attribute__((address_space(1))) * GGlobalIn;
__device__ __forceinline__ void ggoal_x(const GCost& C, int64_t gp, int t, double (&xg)[UN])
{
    if (C.x_path != nullptr) {
        const GGlobalIn q = (GGlobalIn)(C.x_path + gp * C.gsx + (C.path ? (int64_t)t * UN : 0));
#pragma unroll
        for (int a = 0; a < UN; ++a)
            xg[a] = q[a];
    } else {
#pragma unroll
        for (int a = 0; a < UN; ++a)
            xg[a] = C.xg[a];
    }
}
__device__ __forceinline__ void ggoal_u(const GCost& C, int64_t gp, int t, double (&ug)[UM])
{
    if (C.u_path != nullptr) {
        const GGlobalIn q = (GGlobalIn)(C.u_path + gp * C.gsu + (C.path ? (int64_t)t * UM : 0));
#pragma unroll
        for (int a = 0; a < UM; ++a)
            ug[a] = q[a];
    } else {
#pragma unroll
        for (int a = 0; a < UM; ++a)
            ug[a] = C.ug[a];
    }
}
struct GBounds {
    double lo[UM], hi[UM];
};

template <int R>
__device__ __forceinline__ void gload(const double* base, int64_t row0, int64_t b, double* v)
{
#pragma unroll
    for (int i = 0; i < R; ++i)
        v[i] = base[(row0 + i) * 64 + b];
}
template <int R>
__device__ __forceinline__ void gstore(double* base, int64_t row0, int64_t b, const double* v)
{
#pragma unroll
    for (int i = 0; i < R; ++i)
        base[(row0 + i) * 64 + b] = v[i];
}

// QRCost.l, ilqr_cost.py:532-553 (the value carries the 1/2 its gradients lack)
__device__ inline double gcost_l(const GCost& C, const double* x, const double* u, int64_t gp, int t, bool terminal)
{
    double xd[UN], ud[UM], xg[UN];
    ggoal_x(C, gp, t, xg);
    for (int a = 0; a < UN; ++a)
        xd[a] = x[a] - xg[a];
    double sx = qpg_quad(UN, xd, terminal ? C.Qt : C.Q);
    if (!C.path) // PathQRCost.l has no 1/2 (:695-701)
        sx = sx * 0.5;
    if (terminal)
        return sx;
    double ug[UM];
    ggoal_u(C, gp, t, ug);
    for (int a = 0; a < UM; ++a)
        ud[a] = u[a] - ug[a];
    double su = qpg_quad(UM, ud, C.R);
    if (!C.path)
        su = su * 0.5;
    return sx + su;
}
// l_x = x_diff^T (Q + Q^T), ilqr_cost.py:555-569
__device__ inline void gcost_lx(const GCost& C, const double* x, int64_t gp, int t, bool terminal, double* out)
{
    const double* Qm = terminal ? C.Qt : C.Q;
    double xd[UN], xg[UN];
    ggoal_x(C, gp, t, xg);
    for (int a = 0; a < UN; ++a)
        xd[a] = x[a] - xg[a];
    for (int j = 0; j < UN; ++j) {
        double acc = xd[0] * (Qm[j] + Qm[j * UN]);
        for (int a = 1; a < UN; ++a)
            acc = fma(xd[a], Qm[a * UN + j] + Qm[j * UN + a], acc);
        out[j] = acc;
    }
}

struct GAcc {
    double Vx[UN], Vxx[UN * UN];
    double dJ1, dJ2;
    int qp_iter, qp_backtrack, n_clamped;
};

// One step of _backward_pass (ilqr_core.py:127-176) with _compute_Q_terms (:179-228).
__device__ int gbw_step(const UParams& P, const GCost& C, const QPOptsP& QO, const GBounds& bd, const double* x,
                        const double* u, int64_t gp, int t, double mu, GAcc& S, double* ki, double* Ki)
{
    constexpr int n = UN, m = UM;
    double fx[n * n], fu[n * m], lx[n], lu[m];
    double Qx[n], Qu[m], Qxx[n * n], Qux[m * n], Quu[m * m], M1[n * n], M2[m * n], L[m * m], Hff[m * m], T1[n * m];
    const double* Vx = S.Vx;
    const double* Vxx = S.Vxx;
    ilqr_user_jac(P.p, x, u, fx, fu);
    gcost_lx(C, x, gp, t, false, lx);
    {   // l_u = u_diff^T (R + R^T), ilqr_cost.py:571-587
        double ud[m], ug[UM];
        ggoal_u(C, gp, t, ug);
        for (int a = 0; a < m; ++a)
            ud[a] = u[a] - ug[a];
        for (int j = 0; j < m; ++j) {
            double acc = ud[0] * (C.R[j] + C.R[j * m]);
            for (int a = 1; a < m; ++a)
                acc = fma(ud[a], C.R[a * m + j] + C.R[j * m + a], acc);
            lu[j] = acc;
        }
    }
    for (int j = 0; j < n; ++j) {
        double acc = fx[j] * Vx[0];
        for (int a = 1; a < n; ++a)
            acc = fma(fx[a * n + j], Vx[a], acc);
        Qx[j] = lx[j] + acc;
    }
    for (int j = 0; j < m; ++j) {
        double acc = fu[j] * Vx[0];
        for (int a = 1; a < n; ++a)
            acc = fma(fu[a * m + j], Vx[a], acc);
        Qu[j] = lu[j] + acc;
    }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            double acc = fx[a] * Vxx[b];
            for (int c = 1; c < n; ++c)
                acc = fma(fx[c * n + a], Vxx[c * n + b], acc);
            M1[a * n + b] = acc;
        }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            double acc = M1[a * n] * fx[b];
            for (int c = 1; c < n; ++c)
                acc = fma(M1[a * n + c], fx[c * n + b], acc);
            Qxx[a * n + b] = (C.Q[a * n + b] + C.Q[b * n + a]) + acc;
        }
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < n; ++b) { // reg = V_xx + mu I
            double acc = fu[a] * (Vxx[b] + mu * (b == 0 ? 1.0 : 0.0));
            for (int c = 1; c < n; ++c)
                acc = fma(fu[c * m + a], Vxx[c * n + b] + mu * (c == b ? 1.0 : 0.0), acc);
            M2[a * n + b] = acc;
        }
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < n; ++b) {
            double acc = M2[a * n] * fx[b];
            for (int c = 1; c < n; ++c)
                acc = fma(M2[a * n + c], fx[c * n + b], acc);
            Qux[a * n + b] = 0.0 + acc;
        }
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b) {
            double acc = M2[a * n] * fu[b];
            for (int c = 1; c < n; ++c)
                acc = fma(M2[a * n + c], fu[c * m + b], acc);
            Quu[a * m + b] = (C.R[a * m + b] + C.R[b * m + a]) + acc;
        }
    // np.all(np.linalg.eigvals(Q_uu) > 0) else print / warn / break, then cholesky   (:137-141)
    for (int a = 0; a < m * m; ++a)
        if (!(fabs(Quu[a]) < __builtin_inf()))
            return BW_LINALG;
    if (qpg_chol(m, Quu, m, L, m) != 0)
        return BW_NOT_PD;
    // k[i] = cho_solve((R_Q_uu, True), Q_u): the (positive-sign) boxQP warm start   (:145)
    double xq[m], lo[m], hi[m];
    int freev[m];
    for (int a = 0; a < m; ++a) {
        xq[a] = Qu[a];
        lo[a] = bd.lo[a] - u[a];
        hi[a] = bd.hi[a] - u[a];
    }
    qpg_solve(m, L, m, xq);
    QPDenseOut qr;
    boxqp_dense(QO, m, Quu, Qu, lo, hi, true, xq, Hff, freev, qr);
    if (qr.linalg)
        return BW_LINALG;
    S.qp_iter += qr.n_iter;
    S.qp_backtrack += qr.n_bt;
    int nf = 0, fidx[m];
    for (int a = 0; a < m; ++a) {
        ki[a] = xq[a];
        if (freev[a])
            fidx[nf++] = a;
    }
    S.n_clamped += nf < m ? 1 : 0;
    // K[i, free, :] = -cho_solve((H_ff, True), Q_ux[free, :]); clamped rows stay 0   (:157-160)
    for (int a = 0; a < m * n; ++a)
        Ki[a] = 0.0;
    if (nf > 0) {
        double col[m];
        for (int b = 0; b < n; ++b) {
            for (int a = 0; a < nf; ++a)
                col[a] = Qux[fidx[a] * n + b];
            qpg_solve(nf, Hff, m, col);
            for (int a = 0; a < nf; ++a)
                Ki[fidx[a] * n + b] = col[a] * -1;
        }
    }
    // V_x, V_xx (:164-169), Delta_J (:175-176)
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < m; ++b) {
            double acc = Ki[a] * Quu[b];
            for (int c = 1; c < m; ++c)
                acc = fma(Ki[c * n + a], Quu[c * m + b], acc);
            T1[a * m + b] = acc;
        }
    double Vxn[n], Vn[n * n];
    for (int a = 0; a < n; ++a) {
        double t1 = T1[a * m] * ki[0], t2 = Ki[a] * Qu[0], t3 = Qux[a] * ki[0];
        for (int c = 1; c < m; ++c) {
            t1 = fma(T1[a * m + c], ki[c], t1);
            t2 = fma(Ki[c * n + a], Qu[c], t2);
            t3 = fma(Qux[c * n + a], ki[c], t3);
        }
        Vxn[a] = ((Qx[a] + t1) + t2) + t3;
    }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            double t1 = T1[a * m] * Ki[b], t2 = Ki[a] * Qux[b], t3 = Qux[a] * Ki[b];
            for (int c = 1; c < m; ++c) {
                t1 = fma(T1[a * m + c], Ki[c * n + b], t1);
                t2 = fma(Ki[c * n + a], Qux[c * n + b], t2);
                t3 = fma(Qux[c * n + a], Ki[c * n + b], t3);
            }
            Vn[a * n + b] = ((Qxx[a * n + b] + t1) + t2) + t3;
        }
    for (int a = 0; a < n; ++a)
        S.Vx[a] = Vxn[a];
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
            S.Vxx[a * n + b] = 0.5 * (Vn[a * n + b] + Vn[b * n + a]);
    double d1 = ki[0] * Qu[0];
    for (int c = 1; c < m; ++c)
        d1 = fma(ki[c], Qu[c], d1);
    double d2 = 0.0;
    for (int a = 0; a < m; ++a) {
        double acc = Quu[a * m] * ki[0];
        for (int c = 1; c < m; ++c)
            acc = fma(Quu[a * m + c], ki[c], acc);
        d2 = (a == 0) ? ki[0] * acc : fma(ki[a], acc, d2);
    }
    S.dJ1 = S.dJ1 + d1;
    S.dJ2 = S.dJ2 + d2 * 0.5;
    return BW_OK;
}

// One backward sweep; gains of the steps it does not reach are zero (np.zeros_like, :118-119).
__device__ int gbackward(const UParams& P, const GCost& C, const QPOptsP& QO, const GBounds& bd, int T, int64_t b,
                         int64_t gp0, const double* xs, const double* us, double mu, double* k, double* K, GAcc& S,
                         int& stopped_at)
{
    const int64_t gp = gp0 + b; // the problem in column b of the tile: its goal may be its own
    double xT[UN];
    gload<UN>(xs, (int64_t)T * UN, b, xT);
    gcost_lx(C, xT, gp, T, true, S.Vx);
    for (int a = 0; a < UN; ++a)
        for (int c = 0; c < UN; ++c)
            S.Vxx[a * UN + c] = C.Qt[a * UN + c] + C.Qt[c * UN + a];
    S.dJ1 = 0.0;
    S.dJ2 = 0.0;
    S.qp_iter = S.qp_backtrack = S.n_clamped = 0;
    stopped_at = -1;
    int rc = BW_OK;
    for (int i = T - 1; i >= 0; --i) {
        double ki[UM], Ki[UM * UN];
        for (int a = 0; a < UM; ++a)
            ki[a] = 0.0;
        for (int a = 0; a < UM * UN; ++a)
            Ki[a] = 0.0;
        if (rc == BW_OK) {
            double x[UN], u[UM], kt[UM], Kt[UM * UN];
            gload<UN>(xs, (int64_t)i * UN, b, x);
            gload<UM>(us, (int64_t)i * UM, b, u);
            int r = gbw_step(P, C, QO, bd, x, u, gp, i, mu, S, kt, Kt);
            if (r == BW_OK) {
                for (int a = 0; a < UM; ++a)
                    ki[a] = kt[a];
                for (int a = 0; a < UM * UN; ++a)
                    Ki[a] = Kt[a];
            } else {
                rc = r;
                if (r == BW_NOT_PD)
                    stopped_at = i;
            }
        }
        gstore<UM>(k, (int64_t)i * UM, b, ki);
        gstore<UM * UN>(K, (int64_t)i * UM * UN, b, Ki);
    }
    return rc;
}

// _forward_pass + _trajectory_cost (ilqr_core.py:85-107, :230-239); k == nullptr: plain rollout of us.
__device__ double gforward(const UParams& P, const GCost& C, int T, int64_t b, int64_t gp0, const double* xs,
                           const double* us, const double* k, const double* K, double alpha, const double* x0,
                           double* xs_new, double* us_new)
{
    const int64_t gp = gp0 + b;
    double xn[UN], xt[UN];
    gload<UN>(x0, 0, b, xn);
    double J = 0.0;
    for (int t = 0; t < T; ++t) {
        double un[UM];
        gload<UM>(us, (int64_t)t * UM, b, un);
        if (k != nullptr) {
            double xo[UN], dx[UN], kt[UM], Kt[UM * UN];
            gload<UN>(xs, (int64_t)t * UN, b, xo);
            gload<UM>(k, (int64_t)t * UM, b, kt);
            gload<UM * UN>(K, (int64_t)t * UM * UN, b, Kt);
            for (int a = 0; a < UN; ++a)
                dx[a] = xn[a] - xo[a];
            for (int j = 0; j < UM; ++j) {
                double acc = Kt[j * UN] * dx[0];
                for (int a = 1; a < UN; ++a)
                    acc = fma(Kt[j * UN + a], dx[a], acc);
                un[j] = (un[j] + alpha * kt[j]) + acc; // never clamped (:104)
            }
        }
        if (xs_new != nullptr) {
            gstore<UN>(xs_new, (int64_t)t * UN, b, xn);
            gstore<UM>(us_new, (int64_t)t * UM, b, un);
        }
        J = J + gcost_l(C, xn, un, gp, t, false);
        ilqr_user_f(P.p, xn, un, xt);
        for (int a = 0; a < UN; ++a)
            xn[a] = xt[a];
    }
    if (xs_new != nullptr)
        gstore<UN>(xs_new, (int64_t)T * UN, b, xn);
    return J + gcost_l(C, xn, nullptr, gp, T, true);
}

struct UWork {
    double *xs0, *xs1, *us0, *us1, *k0, *k1, *K0, *K1;
};
__device__ __forceinline__ UWork utile(double* ws, int64_t tile, int64_t T)
{
    double* p = ws + tile * utile_rows(T) * 64;
    UWork W;
    W.xs0 = p;
    W.xs1 = W.xs0 + (T + 1) * UN * 64;
    W.us0 = W.xs1 + (T + 1) * UN * 64;
    W.us1 = W.us0 + T * UM * 64;
    W.k0 = W.us1 + T * UM * 64;
    W.k1 = W.k0 + T * UM * 64;
    W.K0 = W.k1 + T * UM * 64;
    W.K1 = W.K0 + T * UM * UN * 64;
    return W;
}

constexpr int GLS_PASS = 8;  // problems per cooperative line-search pass
constexpr int GLS_MAXA = 32; // largest alpha schedule

// iLQR.update_u (ilqr_core.py:241-399), every lane its own problem.
// Line search (:301-313) as in the single-control kernel: every lane rolls out alpha_0 for its own problem;
// the problems that need more hand their remaining candidates to the whole wave (lane = one (problem, alpha)
// pair, costs into LDS) and then scan the costs in the reference's order.  A rollout is a pure function of
// (problem, alpha), so decisions and numbers are those of the serial loop; a rejected iteration costs a few
// rollout times instead of up to 30.
__global__ __launch_bounds__(256, 1) void k_user_solve_m(UParams P, GCost C_arg, SolveOptsP O, GBounds bd, int T, int64_t B,
                                                     double* ws, const double* x0, const double* us_init,
                                                     double* delta_io, double* k_io, double* K_io, double* xs_out,
                                                     double* us_out, double* J_out, double* mu_out,
                                                     int32_t* iters_out, int32_t* status_out, ilqr_trace tr)
{
    // one or four wavefronts per workgroup (the host chooses); a wave only talks to itself
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // the cost matrices are read from LDS (wave-uniform addresses) instead of occupying scalar registers
    __shared__ GCost s_cost;
    __shared__ double s_alpha[GLS_MAXA];
    __shared__ double s_J_all[UWG][GLS_PASS][GLS_MAXA];
    __shared__ int s_lane_all[UWG][GLS_PASS], s_cur_all[UWG][GLS_PASS], s_gcur_all[UWG][GLS_PASS];
    double(&s_J)[GLS_PASS][GLS_MAXA] = s_J_all[wv];
    int(&s_lane)[GLS_PASS] = s_lane_all[wv];
    int(&s_cur)[GLS_PASS] = s_cur_all[wv];
    int(&s_gcur)[GLS_PASS] = s_gcur_all[wv];
    if (threadIdx.x == 0)
        s_cost = C_arg;
    if (threadIdx.x < GLS_MAXA)
        s_alpha[threadIdx.x] = O.alphas[threadIdx.x];
    const GCost& C = s_cost;
    __syncthreads();
    // problems per wavefront: the first O.lanes_per_wave columns of the wave's 64-column tile are in use
    const int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const int64_t gp0 = tile * O.lanes_per_wave; // problem in column 0 of the tile
    const int64_t b = gp0 + lane;
    const bool valid = lane < O.lanes_per_wave && b < B;
    const UWork W = utile(ws, tile, T);
    const int64_t bl = lane;
#define XS(i) ((i) ? W.xs1 : W.xs0)
#define US(i) ((i) ? W.us1 : W.us0)
#define KF(i) ((i) ? W.k1 : W.k0)
#define KG(i) ((i) ? W.K1 : W.K0)
    const QPOptsP& QO = O.qp;
    int cur = 0, gcur = 0;
    double mu = 1.0, delta = 2.0, J = 0.0;
    if (valid) {
        for (int i = 0; i < UN; ++i)
            W.xs0[(int64_t)i * 64 + bl] = x0[b * UN + i];
        for (int t = 0; t < T; ++t) {
            for (int j = 0; j < UM; ++j) {
                W.us0[((int64_t)t * UM + j) * 64 + bl] = us_init[(b * T + t) * UM + j];
                W.k0[((int64_t)t * UM + j) * 64 + bl] = k_io[(b * T + t) * UM + j];
            }
            for (int i = 0; i < UM * UN; ++i)
                W.K0[((int64_t)t * UM * UN + i) * 64 + bl] = K_io[(b * T + t) * UM * UN + i];
        }
        mu = O.mu_from_array ? mu_out[b] : 1.0; // reset per call (:264); delta_ persists (:62)
        delta = delta_io[b];
        // initial rollout and cost (:269-271)
        J = gforward(P, C, T, bl, gp0, nullptr, W.us0, nullptr, nullptr, 0.0, W.xs0, W.xs0, W.us0);
    }
    double dJ1 = 0.0, dJ2 = 0.0, actual = 0.0;
    if (valid && O.mu_from_array && O.dJ_carry != nullptr) { // a continued solve: the sums of its last backward pass
        dJ1 = O.dJ_carry[b];
        dJ2 = O.dJ_carry[B + b];
    }
    int status = ILQR_RAN_OUT, iters = 0;
    bool running = valid && O.max_iter > 0;
    const int n_more = O.n_alpha - 1; // candidates after alpha_0
    for (int it = 0; it < O.max_iter; ++it) {
        if (__ballot(running) == 0ull)
            break;
        GAcc S;
        int stopped_at = -1;
        S.qp_iter = S.qp_backtrack = S.n_clamped = 0;
        int accepted = 0, j = 0;
        double Jnew = 0.0;
        const int nxt = cur ^ 1;
        bool need = false; // the serial loop would go on to alpha_1
        if (running) {
            // backward pass with the LinAlgError retry loop (:287-299)
            for (;;) {
                int gnew = gcur ^ 1;
                int rc = gbackward(P, C, QO, bd, T, bl, gp0, XS(cur), US(cur), mu, KF(gnew), KG(gnew), S, stopped_at);
                if (rc != BW_LINALG) {
                    gcur = gnew;
                    dJ1 = S.dJ1;
                    dJ2 = S.dJ2;
                    break;
                }
                delta = fmax(delta, 1.0) * O.delta0;
                mu = fmax(O.mu_min, mu * delta);
                if (O.mu_max != 0.0 && mu >= O.mu_max)
                    break;
            }
            // line search, alpha_0 (:301-313)
            if (O.n_alpha > 0) {
                double alpha = s_alpha[0];
                Jnew = gforward(P, C, T, bl, gp0, XS(cur), US(cur), KF(gcur), KG(gcur), alpha, XS(cur), XS(nxt), US(nxt));
                double expected = (alpha * dJ1 + alpha * alpha * dJ2) * -1;
                if (expected <= 0.0) {
                    // reject the whole iteration
                } else if ((J - Jnew) / expected > O.scale) {
                    actual = J - Jnew;
                    accepted = 1;
                } else {
                    need = n_more > 0;
                    j = need ? 1 : O.n_alpha;
                }
            }
        }
        // remaining alphas of the problems that need them, GLS_PASS problems per pass
        // (helper lanes read gains another lane of this wave has just stored)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        unsigned long long mask = __ballot(need);
        bool redo = false; // accepted at some alpha_j, j >= 1: its trajectory still has to be written
        while (mask != 0ull) {
            int rank = __popcll(mask & ((1ull << lane) - 1ull));
            bool mine = need && rank < GLS_PASS;
            if (mine) {
                s_lane[rank] = lane;
                s_cur[rank] = cur;
                s_gcur[rank] = gcur;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int np = __popcll(mask);
            np = np < GLS_PASS ? np : GLS_PASS;
            int nc = np * n_more;
            for (int q0 = 0; q0 < nc; q0 += 64) {
                int q = q0 + lane;
                if (q < nc) {
                    int fi = q / n_more, jj = 1 + q % n_more;
                    int64_t pb = s_lane[fi];
                    int pc = s_cur[fi], pg = s_gcur[fi];
                    s_J[fi][jj] = gforward(P, C, T, pb, gp0, XS(pc), US(pc), KF(pg), KG(pg), s_alpha[jj], XS(pc), nullptr, nullptr);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (mine) {
                // the reference's loop from alpha_1 on, on the tabulated costs
                for (j = 1; j < O.n_alpha; ++j) {
                    double alpha = s_alpha[j];
                    Jnew = s_J[rank][j];
                    double expected = (alpha * dJ1 + alpha * alpha * dJ2) * -1;
                    if (expected <= 0.0) {
                        break;
                    } else if ((J - Jnew) / expected > O.scale) {
                        actual = J - Jnew;
                        accepted = 1;
                        redo = true;
                        break;
                    }
                }
                need = false;
            }
            __builtin_amdgcn_wave_barrier();
            mask = __ballot(need);
        }
        if (redo) // the accepted candidate's states and controls (same arithmetic, same bits)
            Jnew = gforward(P, C, T, bl, gp0, XS(cur), US(cur), KF(gcur), KG(gcur), s_alpha[j], XS(cur), XS(nxt), US(nxt));
        if (running) {
            bool stop = false;
            int converged = 0;
            if (accepted) {
                delta = fmin(1.0, delta) / O.delta0;
                mu = mu * delta;
                if (mu <= O.mu_min)
                    mu = 0.0;
                J = Jnew;
                cur = nxt;
                if (actual < O.tol)
                    converged = 1;
            } else {
                delta = fmax(delta, 1.0) * O.delta0;
                mu = fmax(O.mu_min, mu * delta);
                if (O.mu_max != 0.0 && mu >= O.mu_max) {
                    status = ILQR_MU_MAX;
                    stop = true;
                }
            }
            int64_t to = (int64_t)it * B + b;
            if (tr.J) tr.J[to] = J;
            if (tr.accepted) tr.accepted[to] = accepted;
            if (tr.alpha_idx) tr.alpha_idx[to] = j;
            if (tr.mu) tr.mu[to] = mu;
            if (tr.delta) tr.delta[to] = delta;
            if (tr.qp_iter) tr.qp_iter[to] = S.qp_iter;
            if (tr.qp_backtrack) tr.qp_backtrack[to] = S.qp_backtrack;
            if (tr.n_clamped) tr.n_clamped[to] = S.n_clamped;
            if (tr.stopped_at) tr.stopped_at[to] = stopped_at;
            iters = it + 1;
            if (converged)
                status = ILQR_CONVERGED;
            if (stop || converged)
                running = false;
        }
    }
    if (valid) {
        const double* xf = XS(cur);
        const double* uf = US(cur);
        const double* kf = KF(gcur);
        const double* Kf = KG(gcur);
        for (int t = 0; t <= T; ++t)
            for (int i = 0; i < UN; ++i)
                xs_out[(b * (T + 1) + t) * UN + i] = xf[((int64_t)t * UN + i) * 64 + bl];
        for (int t = 0; t < T; ++t) {
            for (int j2 = 0; j2 < UM; ++j2) {
                us_out[(b * T + t) * UM + j2] = uf[((int64_t)t * UM + j2) * 64 + bl];
                k_io[(b * T + t) * UM + j2] = kf[((int64_t)t * UM + j2) * 64 + bl];
            }
            for (int i = 0; i < UM * UN; ++i)
                K_io[(b * T + t) * UM * UN + i] = Kf[((int64_t)t * UM * UN + i) * 64 + bl];
        }
        delta_io[b] = delta;
        J_out[b] = J;
        mu_out[b] = mu;
        iters_out[b] = iters;
        status_out[b] = status;
        if (O.mu_from_array && O.dJ_carry != nullptr) { // only a solve that is being continued carries the sums
            O.dJ_carry[b] = dJ1;
            O.dJ_carry[B + b] = dJ2;
        }
    }
#undef XS
#undef US
#undef KF
#undef KG
}

#endif // ILQR_USER_M

int ucheck(const ilqr_user_model* m)
{
    if (!m)
        return ILQR_ERR_ARG;
    if (m->n != UN || m->m != UM || m->n_params != ILQR_USER_NP)
        return ILQR_ERR_UNSUPPORTED;
    if (m->T < 1 || !m->Q || !m->R || !m->lower || !m->upper || (ILQR_USER_NP > 0 && !m->params))
        return ILQR_ERR_ARG;
    if (m->cost_kind != ILQR_COST_QR && m->cost_kind != ILQR_COST_PATH_QR)
        return ILQR_ERR_UNSUPPORTED;
    if ((m->goal_stride_x != 0) != (m->goal_stride_u != 0))
        return ILQR_ERR_ARG;
    if (m->goal_stride_x != 0 &&
        (!m->x_goal || !m->u_goal ||
         m->goal_stride_x < (m->cost_kind == ILQR_COST_PATH_QR ? ((int64_t)m->T + 1) * UN : UN) ||
         m->goal_stride_u < (m->cost_kind == ILQR_COST_PATH_QR ? (int64_t)m->T * UM : UM)))
        return ILQR_ERR_ARG;
    if (m->cost_kind == ILQR_COST_PATH_QR && (!m->x_goal || !m->u_goal))
        return ILQR_ERR_ARG; // the paths are required (device arrays)
    return ILQR_OK;
}

// Problems per wavefront: half-filled waves while every wave still finds a free SIMD (1024 of them).  A wave
// runs as long as its slowest lane, and an instruction costs the same for 32 lanes as for 64.
int ulanes(int64_t B)
{
    int lpw = B <= 32 * 1024 ? 32 : 64;
    if (const char* e = std::getenv("ILQR_SOLVE_LPW")) {
        int v = std::atoi(e);
        lpw = (v == 16 || v == 32) ? v : 64;
    }
    return lpw;
}
// Wavefronts per workgroup.  The single-control kernel runs 3-5 % faster with four waves on one CU than
// with one wave on each of four CUs (as the built-in cart-pole kernel does).  The multi-control kernel,
// which spills to scratch memory and leans on the CU's vector cache, is 10-18 % slower that way while CUs
// are free; once every CU holds four waves anyway the two shapes measure alike.
int uwaves_per_group(int64_t nwaves)
{
    int w = (UM > 1 && nwaves < 1024) ? 1 : UWG;
    if (const char* e = std::getenv("ILQR_SOLVE_BLOCK")) {
        int v = std::atoi(e);
        if (v == 64 || v == 256)
            w = v / 64;
    }
    return w;
}
int64_t utiles(int64_t B) { int lpw = ulanes(B); return (B + lpw - 1) / lpw; }
int64_t uld(int64_t B) { return utiles(B) * 64; }

} // namespace

extern "C" {

int ilqr_user_dims(int32_t* n, int32_t* n_params)
{
    if (n)
        *n = UN;
    if (n_params)
        *n_params = ILQR_USER_NP;
    return ILQR_OK;
}

int ilqr_user_controls(void) { return UM; }

size_t ilqr_user_workspace_bytes(const ilqr_user_model* model, int64_t B)
{
    if (ucheck(model) != ILQR_OK || B < 1)
        return 0;
    int64_t T = model->T;
    return (size_t)uld(B) * sizeof(double) * (size_t)utile_rows(T);
}

int ilqr_user_solve_batched(const ilqr_user_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                            const double* us_init, double* delta, double* k, double* K, double* xs, double* us,
                            double* J, double* mu, int32_t* iters, int32_t* status, const ilqr_trace* trace,
                            void* workspace, size_t workspace_bytes, void* stream)
{
    int rc = ucheck(model);
    if (rc != ILQR_OK)
        return rc;
    if (!opts || !opts->alphas || opts->n_alpha < 0 || opts->n_alpha > 32 || opts->max_iter < 0 ||
        opts->qp_max_iter < 0 || !(opts->qp_step_dec > 0.0) || !(opts->qp_step_dec < 1.0) || !(opts->qp_min_step > 0.0))
        return ILQR_ERR_ARG;
    if (B < 1 || !x0 || !us_init || !delta || !k || !K || !xs || !us || !J || !mu || !iters || !status || !workspace)
        return ILQR_ERR_ARG;
    if (workspace_bytes < ilqr_user_workspace_bytes(model, B))
        return ILQR_ERR_WORKSPACE;
    UParams P;
    for (int i = 0; i < UNP; ++i)
        P.p[i] = i < ILQR_USER_NP ? model->params[i] : 0.0;
    const double* Qt = model->Q_terminal ? model->Q_terminal : model->Q;
    const bool path = model->cost_kind == ILQR_COST_PATH_QR;
#if ILQR_USER_M == 1
    UCost C;
    for (int a = 0; a < UN; ++a) {
        for (int b = 0; b < UN; ++b) {
            C.Q[a * UN + b] = model->Q[a * UN + b];
            C.Qt[a * UN + b] = Qt[a * UN + b];
            C.QQ[a * UN + b] = model->Q[a * UN + b] + model->Q[b * UN + a];
            C.QQt[a * UN + b] = Qt[a * UN + b] + Qt[b * UN + a];
        }
        C.xg[a] = (!path && model->goal_stride_x == 0 && model->x_goal) ? model->x_goal[a] : 0.0;
    }
    C.R = model->R[0];
    C.RR = model->R[0] + model->R[0];
    C.ug = (!path && model->goal_stride_x == 0 && model->u_goal) ? model->u_goal[0] : 0.0;
    const bool dev_goals = path || model->goal_stride_x != 0;
    C.path = path ? 1 : 0;
    C.x_path = dev_goals ? model->x_goal : nullptr;
    C.u_path = dev_goals ? model->u_goal : nullptr;
    C.gsx = model->goal_stride_x;
    C.gsu = model->goal_stride_u;
    UBounds bd;
    bd.lo = model->lower[0];
    bd.hi = model->upper[0];
#else
    GCost C;
    for (int a = 0; a < UN * UN; ++a) {
        C.Q[a] = model->Q[a];
        C.Qt[a] = Qt[a];
    }
    for (int a = 0; a < UM * UM; ++a)
        C.R[a] = model->R[a];
    const bool dev_goals = path || model->goal_stride_x != 0;
    for (int a = 0; a < UN; ++a)
        C.xg[a] = (!dev_goals && model->x_goal) ? model->x_goal[a] : 0.0;
    C.path = path ? 1 : 0;
    C.x_path = dev_goals ? model->x_goal : nullptr;
    C.u_path = dev_goals ? model->u_goal : nullptr;
    C.gsx = model->goal_stride_x;
    C.gsu = model->goal_stride_u;
    GBounds bd;
    for (int a = 0; a < UM; ++a) {
        C.ug[a] = (!dev_goals && model->u_goal) ? model->u_goal[a] : 0.0;
        bd.lo[a] = model->lower[a];
        bd.hi[a] = model->upper[a];
    }
#endif
    SolveOptsP O;
    O.max_iter = opts->max_iter;
    O.n_alpha = opts->n_alpha;
    O.mu_from_array = (opts->mu_init >= 0.0) ? 1 : 0;
    O.lanes_per_wave = ulanes(B);
    O.tol = opts->tol;
    O.scale = opts->scale;
    O.mu_min = opts->mu_min;
    O.mu_max = opts->mu_max;
    O.delta0 = opts->delta0;
    O.dJ_carry = opts->dJ_carry;
    for (int i = 0; i < 32; ++i)
        O.alphas[i] = i < opts->n_alpha ? opts->alphas[i] : 0.0;
    O.qp.max_iter = opts->qp_max_iter;
    O.qp.min_grad = opts->qp_min_grad;
    O.qp.min_rel = opts->qp_min_rel_improve;
    O.qp.step_dec = opts->qp_step_dec;
    O.qp.min_step = opts->qp_min_step;
    O.qp.armijo = opts->qp_armijo;
    {
        volatile double step = 1.0;
        int n = 0;
        do {
            step = step * opts->qp_step_dec;
            ++n;
        } while (!(step < opts->qp_min_step) && n < 1000000);
        O.qp.n_exhaust = n;
        O.qp.step_last = step;
        O.qp.tab_n = n < QP_TAB ? n : QP_TAB;
        O.qp.inv_log2_dec = (float)(1.0 / std::log2(opts->qp_step_dec));
    }
    int64_t T = model->T;
    ilqr_trace tr;
    if (trace)
        tr = *trace;
    else
        std::memset(&tr, 0, sizeof(tr));
#if ILQR_USER_M > 1
#define k_user_solve k_user_solve_m
#endif
    const int wpg = uwaves_per_group(utiles(B));
    hipLaunchKernelGGL(k_user_solve, dim3((unsigned)((utiles(B) + wpg - 1) / wpg)), dim3(64 * wpg), 0, reinterpret_cast<hipStream_t>(stream),
                       P, C, O, bd, (int)T, B, static_cast<double*>(workspace), x0, us_init, delta, k, K, xs, us, J, mu,
                       iters, status, tr);
    return hipGetLastError() == hipSuccess ? ILQR_OK : ILQR_ERR_HIP;
}

} // extern "C"
