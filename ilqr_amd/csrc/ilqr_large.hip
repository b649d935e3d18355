// Large-state path of the batched control-limited iLQR solve: n = 32 states, m = 8 controls,
// one PROBLEM PER WAVEFRONT.  Every matrix product of the backward pass and of the rollouts is
// an fp64 MFMA (v_mfma_f64_16x16x4_f64); the line search rolls out 16 step sizes at once as
// the 16 columns of a 32 x 16 state matrix.
//
// Arithmetic contract.  v_mfma_f64_16x16x4_f64 computes, per output element, the k-ascending
// chain D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C)))) (checked bit for bit on
// gfx950), so a product accumulated over consecutive k-blocks from C = 0 is exactly the
// left-to-right fma chain "first product, then fma" that the CPU restatement of the reference's
// np.dot calls uses; only the sign of an exact zero can differ.  Everything else (additions of
// the cost terms, the dynamics non-linearity, Cholesky, boxQP) is plain unfused fp64 in the
// order of the reference (compile with -ffp-contract=off).
//
// Register layouts (lane l, g = l >> 4, c = l & 15, r = 0..3):
//   N layout of a matrix X, 16 x 16 tiles: tile (rt, ct), register r holds X[16 rt + g + 4 r][16 ct + c].
//     It is both the C/D layout of the MFMA and its B-operand layout for the four k-blocks of a
//     row tile, and the A-operand layout of X^T.  Hence every product is written  O = X^T Y
//     with X, Y, O in N layout and no data movement between products.
//   V layout of a vector v: register (rt, r) holds v[16 rt + g + 4 r], replicated over c.  A
//     matrix-vector product X^T v with v in V layout returns its result in V layout.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/ilqr_hip.h"

namespace {

constexpr int LN = 32; // states
constexpr int LM = 8;  // controls
constexpr int LA = 16; // step sizes rolled out together

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 zero4() { return d4{0.0, 0.0, 0.0, 0.0}; }

__device__ __forceinline__ d4 mfma4(double a, double b, d4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// O[i][j] = sum over the KT row tiles (RK k-blocks of 4 rows each) of X^T Y.
template <int KT, int MI, int NJ, int RK = 4>
__device__ __forceinline__ void mm_tn(const d4 (&X)[KT][MI], const d4 (&Y)[KT][NJ], d4 (&O)[MI][NJ])
{
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            d4 acc = zero4();
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < RK; ++r)
                    acc = mfma4(X[kt][i][r], Y[kt][j][r], acc);
            O[i][j] = acc;
        }
}

struct LaneId {
    int g, c;
};
__device__ __forceinline__ LaneId lane_id()
{
    int l = threadIdx.x & 63;
    return LaneId{l >> 4, l & 15};
}

// N layout of a row-major matrix P[rows][ld] (rows x cols valid, zero beyond)
__device__ __forceinline__ d4 load_n(const double* P, int ld, int rows, int cols, int rt, int ct, LaneId L)
{
    d4 v;
    int col = 16 * ct + L.c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int row = 16 * rt + L.g + 4 * r;
        v[r] = (row < rows && col < cols) ? P[row * ld + col] : 0.0;
    }
    return v;
}
// N layout of P^T for a row-major P[prow][ld]: element [16 rt + g + 4 r][16 ct + c] = P[16 ct + c][16 rt + g + 4 r]
__device__ __forceinline__ d4 load_nt(const double* P, int ld, int prows, int pcols, int rt, int ct, LaneId L)
{
    d4 v;
    int prow = 16 * ct + L.c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int pcol = 16 * rt + L.g + 4 * r;
        v[r] = (prow < prows && pcol < pcols) ? P[prow * ld + pcol] : 0.0;
    }
    return v;
}
// V layout of a vector of `len` entries
__device__ __forceinline__ d4 load_v(const double* p, int len, int rt, LaneId L)
{
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int i = 16 * rt + L.g + 4 * r;
        v[r] = i < len ? p[i] : 0.0;
    }
    return v;
}
__device__ __forceinline__ void store_v(double* p, int len, int rt, d4 v, LaneId L)
{
    if (L.c == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int i = 16 * rt + L.g + 4 * r;
            if (i < len)
                p[i] = v[r];
        }
    }
}
__device__ __forceinline__ void store_n(double* P, int ld, int rows, int cols, int rt, int ct, d4 v, LaneId L)
{
    int col = 16 * ct + L.c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int row = 16 * rt + L.g + 4 * r;
        if (row < rows && col < cols)
            P[row * ld + col] = v[r];
    }
}

struct LargeP { // kernel arguments: the model, all device pointers
    const double *A, *B, *W, *Q, *R, *Qt, *xg, *ug, *lower, *upper;
    double dt, sig_c;
    int T;
};

struct LQPOpts {
    int max_iter;
    double min_grad, min_rel, step_dec, min_step, armijo;
};

struct LSolveOpts {
    int max_iter, n_alpha, mu_from_array;
    double tol, scale, mu_min, mu_max, delta0;
    double alphas[32];
    LQPOpts qp;
};

// ------------------------------------------------------------------ boxQP in LDS
//
// boxQP (ilqr_boxQP.py:9-186) for m = 8.  The state lives in LDS; every lane of the wave
// executes the same scalar program on it (uniform control flow, broadcast reads, identical
// stores), which keeps the 8 x 8 factorisations and the free-subspace bookkeeping a
// transliteration of the reference.
struct QPMem {
    double H[LM * LM], g[LM], lo[LM], hi[LM], x[LM];
    double Hff[LM * LM], Lfull[LM * LM], sub[LM * LM];
    double grad[LM], search[LM], xc[LM], rhs[LM];
    int clamped[LM], freev[LM], oldclamped[LM], fidx[LM], cidx[LM];
};

__device__ __forceinline__ double lclamp(double lo, double hi, double v)
{
    double t = (v != v) ? v : (hi < v ? hi : v);
    return (t != t) ? t : (lo > t ? lo : t);
}

__device__ double lq_quad(int m, const double* v, const double* M)
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
__device__ double lq_value(int m, const double* H, const double* g, const double* x)
{
    double xg = x[0] * g[0];
    for (int i = 1; i < m; ++i)
        xg = fma(x[i], g[i], xg);
    return xg + 0.5 * lq_quad(m, x, H);
}
__device__ int lq_chol(int d, const double* A, int lda, double* L, int ldl)
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
__device__ void lq_solve(int d, const double* L, int ldl, double* b)
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

struct LQPOut {
    int result, hd, nf, n_iter, n_factor, n_bt;
    bool linalg;
};

// H, g, lo, hi and the warm start (in Q.x) are in place.  Returns through Q.x, Q.Hff, Q.freev, Q.fidx.
__device__ void lq_boxqp(const LQPOpts& O, QPMem& Q, LQPOut& out)
{
    const int m = LM;
    for (int i = 0; i < m * m; ++i)
        Q.Hff[i] = 1.0;
    bool h_finite = true;
    for (int i = 0; i < m * m; ++i)
        h_finite = h_finite && (fabs(Q.H[i]) < __builtin_inf());
    for (int i = 0; i < m; ++i) {
        Q.clamped[i] = 0;
        Q.freev[i] = 0;
        Q.x[i] = lclamp(Q.lo[i], Q.hi[i], Q.x[i]);
    }
    double value = lq_value(m, Q.H, Q.g, Q.x), oldvalue = 0.0;
    int result = -1, hd = m, n_iter = 0, n_factor = 0, n_bt = 0, nf = 0;
    bool h_pd = h_finite && lq_chol(m, Q.H, m, Q.Lfull, m) == 0;
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
            double acc = Q.H[i * m] * Q.x[0];
            for (int j = 1; j < m; ++j)
                acc = fma(Q.H[i * m + j], Q.x[j], acc);
            Q.grad[i] = Q.g[i] + acc;
        }
        int nclamped = 0;
        for (int i = 0; i < m; ++i) {
            Q.oldclamped[i] = Q.clamped[i];
            int cl = ((Q.x[i] == Q.lo[i]) && (Q.grad[i] > 0.0)) || ((Q.x[i] == Q.hi[i]) && (Q.grad[i] < 0.0));
            Q.clamped[i] = cl;
            Q.freev[i] = !cl;
            nclamped += cl;
        }
        if (nclamped == m) {
            result = 6;
            break;
        }
        if (!h_finite) {
            linalg = true;
            break;
        }
        if (!h_pd) {
            result = -1;
            break;
        }
        bool factorize = iter == 1;
        for (int i = 0; i < m; ++i)
            factorize = factorize || (Q.oldclamped[i] != Q.clamped[i]);
        int nc = 0;
        nf = 0;
        for (int i = 0; i < m; ++i) {
            if (Q.freev[i])
                Q.fidx[nf++] = i;
            else
                Q.cidx[nc++] = i;
        }
        if (factorize) {
            for (int a = 0; a < nf; ++a)
                for (int b = 0; b < nf; ++b)
                    Q.sub[a * m + b] = Q.H[Q.fidx[a] * m + Q.fidx[b]];
            if (lq_chol(nf, Q.sub, m, Q.Hff, m) != 0) {
                linalg = true;
                break;
            }
            hd = nf;
            ++n_factor;
        }
        double ss = 0.0;
        for (int a = 0; a < nf; ++a) {
            double ga = Q.grad[Q.fidx[a]];
            ss = (a == 0) ? ga * ga : fma(ga, ga, ss);
        }
        double gnorm = nf == 1 ? fabs(Q.grad[Q.fidx[0]]) : sqrt(ss);
        if (gnorm < O.min_grad) {
            result = 5;
            break;
        }
        ++n_iter;
        for (int a = 0; a < nf; ++a) {
            double acc = 0.0;
            for (int b = 0; b < nc; ++b) {
                double h = Q.H[Q.fidx[a] * m + Q.cidx[b]], xv = Q.x[Q.cidx[b]];
                acc = (b == 0) ? h * xv : fma(h, xv, acc);
            }
            Q.rhs[a] = Q.g[Q.fidx[a]] + acc;
        }
        lq_solve(nf, Q.Hff, m, Q.rhs);
        for (int i = 0; i < m; ++i)
            Q.search[i] = 0.0;
        for (int a = 0; a < nf; ++a)
            Q.search[Q.fidx[a]] = Q.rhs[a] * -1 - Q.x[Q.fidx[a]];

        double step = 1.0;
        for (int i = 0; i < m; ++i)
            Q.xc[i] = lclamp(Q.lo[i], Q.hi[i], Q.x[i] + step * Q.search[i]);
        double sdotg = (Q.xc[0] - Q.x[0]) * Q.grad[0];
        for (int i = 1; i < m; ++i)
            sdotg = sdotg + (Q.xc[i] - Q.x[i]) * Q.grad[i];
        double vc = lq_value(m, Q.H, Q.g, Q.xc);
        while ((vc - oldvalue) / (sdotg + 1e-10) < O.armijo) {
            step = step * O.step_dec;
            ++n_bt;
            for (int i = 0; i < m; ++i)
                Q.xc[i] = lclamp(Q.lo[i], Q.hi[i], Q.x[i] + step * Q.search[i]);
            vc = lq_value(m, Q.H, Q.g, Q.xc);
            sdotg = (Q.xc[0] - Q.x[0]) * Q.grad[0];
            for (int i = 1; i < m; ++i)
                sdotg = sdotg + (Q.xc[i] - Q.x[i]) * Q.grad[i];
            if (step < O.min_step) {
                result = 2;
                break;
            }
        }
        for (int i = 0; i < m; ++i)
            Q.x[i] = Q.xc[i];
        value = vc;
    }
    // the free set the caller uses is the one of the last classification
    nf = 0;
    for (int i = 0; i < m; ++i)
        if (Q.freev[i])
            Q.fidx[nf++] = i;
    out.result = result;
    out.hd = hd;
    out.nf = nf;
    out.n_iter = n_iter;
    out.n_factor = n_factor;
    out.n_bt = n_bt;
    out.linalg = linalg;
}

// ------------------------------------------------------------------ per-wave LDS

struct WaveMem {
    QPMem qp;
    double Quu[LM * LM], Qu[LM], kq[LM];
    double Qux[LM][LN + 1];
    double K[LM][LN + 1];
    double Vn[LN][LN + 1];
    double J[LA];
};

// ------------------------------------------------------------------ backward pass

struct BwOut {
    double dJ1, dJ2;
    int stopped_at, qp_iter, qp_backtrack, n_clamped, rc; // rc: 0 ok, 2 LinAlgError
};

// _backward_pass + _compute_Q_terms (ilqr_core.py:109-228) for one problem.
// xs [T+1][32], us [T][8] -> k [T][8], K [T][8][32] (zeros where the sweep does not reach).
__device__ void large_backward(const LargeP& P, const LQPOpts& QO, WaveMem& S, const double* xs, const double* us,
                               double mu, double* k, double* K, BwOut& out, int32_t* qp_result, int32_t* free_mask)
{
    const LaneId L = lane_id();
    const int T = P.T;
    const int lane = threadIdx.x & 63;
    // zero-fill the gains and the optional records (np.zeros_like, ilqr_core.py:118-119)
    for (int i = lane; i < T * LM; i += 64)
        k[i] = 0.0;
    for (int i = lane; i < T * LM * LN; i += 64)
        K[i] = 0.0;
    if (qp_result)
        for (int i = lane; i < T; i += 64)
            qp_result[i] = -100;
    if (free_mask)
        for (int i = lane; i < T * LM; i += 64)
            free_mask[i] = 0;
    __syncthreads();

    // constants in N layout
    d4 QQ[2][2], QQt[2][2], fu[2][1], RR[1][1];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            QQ[rt][ct] = load_n(P.Q, LN, LN, LN, rt, ct, L) + load_nt(P.Q, LN, LN, LN, rt, ct, L);
            QQt[rt][ct] = load_n(P.Qt, LN, LN, LN, rt, ct, L) + load_nt(P.Qt, LN, LN, LN, rt, ct, L);
        }
        fu[rt][0] = P.dt * load_n(P.B, LM, LN, LM, rt, 0, L); // f_u = dt B
    }
    RR[0][0] = load_n(P.R, LM, LM, LM, 0, 0, L) + load_nt(P.R, LM, LM, LM, 0, 0, L);
    d4 xg[2][1], ug[1][1];
    xg[0][0] = load_v(P.xg, LN, 0, L);
    xg[1][0] = load_v(P.xg, LN, 1, L);
    ug[0][0] = load_v(P.ug, LM, 0, L);
    d4 lowv = load_v(P.lower, LM, 0, L), upv = load_v(P.upper, LM, 0, L);

    // V_x, V_xx = terminal l_x, l_xx (:120-121)
    d4 Vx[2][1], Vxx[2][2];
    {
        d4 xd[2][1];
        xd[0][0] = load_v(xs + (size_t)T * LN, LN, 0, L) - xg[0][0];
        xd[1][0] = load_v(xs + (size_t)T * LN, LN, 1, L) - xg[1][0];
        mm_tn<2, 2, 1>(QQt, xd, Vx); // l_x[j] = sum_a xd[a] (Q + Q^T)[a][j]
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                Vxx[rt][ct] = QQt[rt][ct];
    }
    out.dJ1 = 0.0;
    out.dJ2 = 0.0;
    out.stopped_at = -1;
    out.qp_iter = out.qp_backtrack = out.n_clamped = 0;
    out.rc = 0;

    for (int i = T - 1; i >= 0; --i) {
        d4 xv[2][1], uv[1][1];
        xv[0][0] = load_v(xs + (size_t)i * LN, LN, 0, L);
        xv[1][0] = load_v(xs + (size_t)i * LN, LN, 1, L);
        uv[0][0] = load_v(us + (size_t)i * LM, LM, 0, L);

        // z = W x, g_i = c phi'(z_i), f_x = I + dt (A + diag(g) W)
        d4 fx[2][2];
        {
            d4 WT[2][2], z[2][1];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    WT[rt][ct] = load_nt(P.W, LN, LN, LN, rt, ct, L);
            mm_tn<2, 2, 1>(WT, xv, z);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                d4 gi;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double zz = z[rt][0][r];
                    double s = 1.0 + zz * zz;
                    gi[r] = P.sig_c * (1.0 / (s * sqrt(s)));
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    d4 An = load_n(P.A, LN, LN, LN, rt, ct, L), Wn = load_n(P.W, LN, LN, LN, rt, ct, L);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double id = (16 * rt + L.g + 4 * r == 16 * ct + L.c) ? 1.0 : 0.0;
                        fx[rt][ct][r] = id + P.dt * (An[r] + gi[r] * Wn[r]);
                    }
                }
            }
        }
        // l_x, l_u (ilqr_cost.py:555-587)
        d4 lx[2][1], lu[1][1];
        {
            d4 xd[2][1], ud[1][1];
            xd[0][0] = xv[0][0] - xg[0][0];
            xd[1][0] = xv[1][0] - xg[1][0];
            ud[0][0] = uv[0][0] - ug[0][0];
            mm_tn<2, 2, 1>(QQ, xd, lx);
            mm_tn<1, 1, 1, 2>(RR, ud, lu);
        }
        // Q_x = l_x + f_x^T V_x ; Q_u = l_u + f_u^T V_x          (:206-207)
        d4 Qx[2][1], Qu[1][1];
        mm_tn<2, 2, 1>(fx, Vx, Qx);
        mm_tn<2, 1, 1>(fu, Vx, Qu);
        Qx[0][0] = lx[0][0] + Qx[0][0];
        Qx[1][0] = lx[1][0] + Qx[1][0];
        Qu[0][0] = lu[0][0] + Qu[0][0];
        // M1^T = V_xx^T f_x ; Q_xx = l_xx + M1 f_x                (:208)
        d4 Qxx[2][2];
        {
            d4 M1T[2][2];
            mm_tn<2, 2, 2>(Vxx, fx, M1T);
            mm_tn<2, 2, 2>(M1T, fx, Qxx);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    Qxx[rt][ct] = QQ[rt][ct] + Qxx[rt][ct];
        }
        // reg = V_xx + mu I ; M2^T = reg^T f_u ; Q_ux = M2 f_x ; Q_uu = l_uu + M2 f_u   (:210-212)
        d4 Qux[1][2], Quu[1][1];
        {
            d4 reg[2][2], M2T[2][1];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double id = (16 * rt + L.g + 4 * r == 16 * ct + L.c) ? 1.0 : 0.0;
                        reg[rt][ct][r] = Vxx[rt][ct][r] + mu * id;
                    }
            mm_tn<2, 2, 1>(reg, fu, M2T);
            mm_tn<2, 1, 2>(M2T, fx, Qux);
            mm_tn<2, 1, 1>(M2T, fu, Quu);
            Qux[0][0] = 0.0 + Qux[0][0];
            Qux[0][1] = 0.0 + Qux[0][1];
            Quu[0][0] = RR[0][0] + Quu[0][0];
        }
        // ---- hand Q_uu, Q_u, Q_ux and the bounds to the scalar part through LDS
        store_n(S.qp.H, LM, LM, LM, 0, 0, Quu[0][0], L);
        store_v(S.qp.g, LM, 0, Qu[0][0], L);
        store_v(S.qp.lo, LM, 0, lowv - uv[0][0], L);
        store_v(S.qp.hi, LM, 0, upv - uv[0][0], L);
        store_n(&S.Qux[0][0], LN + 1, LM, LN, 0, 0, Qux[0][0], L);
        store_n(&S.Qux[0][0], LN + 1, LM, LN, 0, 1, Qux[0][1], L);
        __syncthreads();
        // np.all(np.linalg.eigvals(Q_uu) > 0) else print / warn / break   (:137-140)
        bool finite = true;
        for (int a = 0; a < LM * LM; ++a)
            finite = finite && (fabs(S.qp.H[a]) < __builtin_inf());
        if (!finite) {
            out.rc = 2;
            break;
        }
        if (lq_chol(LM, S.qp.H, LM, S.qp.Lfull, LM) != 0) {
            out.stopped_at = i;
            break;
        }
        // k[i] = cho_solve((R_Q_uu, True), Q_u): the (positive-sign) boxQP warm start   (:145)
        for (int a = 0; a < LM; ++a)
            S.qp.x[a] = S.qp.g[a];
        lq_solve(LM, S.qp.Lfull, LM, S.qp.x);
        LQPOut qo;
        lq_boxqp(QO, S.qp, qo);
        if (qo.linalg) {
            out.rc = 2;
            break;
        }
        out.qp_iter += qo.n_iter;
        out.qp_backtrack += qo.n_bt;
        out.n_clamped += qo.nf < LM ? 1 : 0;
        if (qp_result && lane == 0)
            qp_result[i] = qo.result;
        if (free_mask && lane < LM)
            free_mask[(size_t)i * LM + lane] = S.qp.freev[lane];
        // K[i, free, :] = -cho_solve((H_ff, True), Q_ux[free, :]); clamped rows stay 0   (:157-160)
        if (lane < LN) {
            double col[LM];
#pragma unroll
            for (int a = 0; a < LM; ++a)
                col[a] = a < qo.nf ? S.Qux[S.qp.fidx[a < qo.nf ? a : 0]][lane] : 0.0;
            // forward and back substitution with the nf x nf factor (row stride 8)
#pragma unroll
            for (int a = 0; a < LM; ++a)
                if (a < qo.nf) {
                    double s = col[a];
#pragma unroll
                    for (int b = 0; b < LM; ++b)
                        if (b < a)
                            s = fma(-S.qp.Hff[a * LM + b], col[b], s);
                    col[a] = s * (1.0 / S.qp.Hff[a * LM + a]);
                }
#pragma unroll
            for (int a = LM - 1; a >= 0; --a)
                if (a < qo.nf) {
                    double s = col[a];
#pragma unroll
                    for (int b = 0; b < LM; ++b)
                        if (b > a && b < qo.nf)
                            s = fma(-S.qp.Hff[b * LM + a], col[b], s);
                    col[a] = s * (1.0 / S.qp.Hff[a * LM + a]);
                }
#pragma unroll
            for (int a = 0; a < LM; ++a)
                S.K[a][lane] = 0.0;
#pragma unroll
            for (int a = 0; a < LM; ++a)
                if (a < qo.nf)
                    S.K[S.qp.fidx[a]][lane] = col[a] * -1;
        }
        __syncthreads();
        d4 Kn[1][2], kv[1][1];
        Kn[0][0] = load_n(&S.K[0][0], LN + 1, LM, LN, 0, 0, L);
        Kn[0][1] = load_n(&S.K[0][0], LN + 1, LM, LN, 0, 1, L);
        kv[0][0] = load_v(S.qp.x, LM, 0, L);
        store_n(K + (size_t)i * LM * LN, LN, LM, LN, 0, 0, Kn[0][0], L);
        store_n(K + (size_t)i * LM * LN, LN, LM, LN, 0, 1, Kn[0][1], L);
        store_v(k + (size_t)i * LM, LM, 0, kv[0][0], L);

        // V_x = Q_x + K^T Q_uu k + K^T Q_u + Q_ux^T k                    (:164)
        // V_xx = Q_xx + K^T Q_uu K + K^T Q_ux + Q_ux^T K, symmetrised    (:167-169)
        {
            d4 T1T[1][2];
            mm_tn<1, 1, 2, 2>(Quu, Kn, T1T); // (K^T Q_uu)^T
            d4 t1[2][2], t2[2][2], t3[2][2], t1v[2][1], t2v[2][1], t3v[2][1];
            mm_tn<1, 2, 2, 2>(T1T, Kn, t1);
            mm_tn<1, 2, 2, 2>(Kn, Qux, t2);
            mm_tn<1, 2, 2, 2>(Qux, Kn, t3);
            mm_tn<1, 2, 1, 2>(T1T, kv, t1v);
            mm_tn<1, 2, 1, 2>(Kn, Qu, t2v);
            mm_tn<1, 2, 1, 2>(Qux, kv, t3v);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                Vx[rt][0] = ((Qx[rt][0] + t1v[rt][0]) + t2v[rt][0]) + t3v[rt][0];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    d4 vn = ((Qxx[rt][ct] + t1[rt][ct]) + t2[rt][ct]) + t3[rt][ct];
                    Vxx[rt][ct] = vn;
                    store_n(&S.Vn[0][0], LN + 1, LN, LN, rt, ct, vn, L);
                }
            }
            __syncthreads();
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    d4 vt = load_nt(&S.Vn[0][0], LN + 1, LN, LN, rt, ct, L);
                    Vxx[rt][ct] = 0.5 * (Vxx[rt][ct] + vt);
                }
        }
        // Delta_J1 += k.Q_u ; Delta_J2 += 0.5 k.(Q_uu k)                 (:175-176)
        {
            const double* kq = S.qp.x;
            double d1 = kq[0] * S.qp.g[0];
            for (int c = 1; c < LM; ++c)
                d1 = fma(kq[c], S.qp.g[c], d1);
            double d2 = 0.0;
            for (int a = 0; a < LM; ++a) {
                double acc = S.qp.H[a * LM] * kq[0];
                for (int c = 1; c < LM; ++c)
                    acc = fma(S.qp.H[a * LM + c], kq[c], acc);
                d2 = (a == 0) ? kq[0] * acc : fma(kq[a], acc, d2);
            }
            out.dJ1 = out.dJ1 + d1;
            out.dJ2 = out.dJ2 + d2 * 0.5;
        }
        __syncthreads(); // the LDS images are rewritten by the next step
    }
}

// ------------------------------------------------------------------ rollouts

// _forward_pass + _trajectory_cost (ilqr_core.py:85-107, :230-239) for 16 step sizes at once
// (column c of every matrix belongs to alpha[c]); with ROLL the controls are us themselves
// (_dynamics_rollout, :69-83).  J of column c ends up in S.J[c].  The trajectory of column
// `store_col` (if >= 0) is written to xs_new / us_new.
template <bool ROLL>
__device__ void large_forward(const LargeP& P, WaveMem& S, const double* xs, const double* us, const double* k,
                              const double* K, const double* x0, double alpha_c, int store_col, double* xs_new,
                              double* us_new)
{
    const LaneId L = lane_id();
    const int T = P.T;
    d4 Qn[2][2], Qtn[2][2], Rn[1][1], WT[2][2], AT[2][2], BT[1][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            Qn[rt][ct] = load_n(P.Q, LN, LN, LN, rt, ct, L);
            Qtn[rt][ct] = load_n(P.Qt, LN, LN, LN, rt, ct, L);
            WT[rt][ct] = load_nt(P.W, LN, LN, LN, rt, ct, L);
            AT[rt][ct] = load_nt(P.A, LN, LN, LN, rt, ct, L);
        }
    Rn[0][0] = load_n(P.R, LM, LM, LM, 0, 0, L);
    BT[0][0] = load_nt(P.B, LM, LN, LM, 0, 0, L); // B^T [8][32]
    BT[0][1] = load_nt(P.B, LM, LN, LM, 0, 1, L);
    d4 xg[2], ug = load_v(P.ug, LM, 0, L);
    xg[0] = load_v(P.xg, LN, 0, L);
    xg[1] = load_v(P.xg, LN, 1, L);
    const bool diag_lane[4] = {L.g == L.c, L.g + 4 == L.c, L.g + 8 == L.c, L.g + 12 == L.c};

    d4 Xh[2][1];
    Xh[0][0] = load_v(x0, LN, 0, L);
    Xh[1][0] = load_v(x0, LN, 1, L);
    double J = 0.0; // meaningful on the diagonal lanes (row == column) of tile (0, 0)
    for (int t = 0; t < T; ++t) {
        d4 Uh[1][1];
        d4 uv = load_v(us + (size_t)t * LM, LM, 0, L);
        if (ROLL) {
            Uh[0][0] = uv;
        } else {
            // u_new = u + alpha k + K (x_new - x)   (:104, never clamped)
            d4 dX[2][1], KT[2][1], acc[1][1];
            dX[0][0] = Xh[0][0] - load_v(xs + (size_t)t * LN, LN, 0, L);
            dX[1][0] = Xh[1][0] - load_v(xs + (size_t)t * LN, LN, 1, L);
            KT[0][0] = load_nt(K + (size_t)t * LM * LN, LN, LM, LN, 0, 0, L);
            KT[1][0] = load_nt(K + (size_t)t * LM * LN, LN, LM, LN, 1, 0, L);
            mm_tn<2, 1, 1>(KT, dX, acc);
            d4 kv = load_v(k + (size_t)t * LM, LM, 0, L);
            Uh[0][0] = (uv + alpha_c * kv) + acc[0][0];
        }
        if (store_col >= 0 && L.c == store_col) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xs_new[(size_t)t * LN + 16 * rt + L.g + 4 * r] = Xh[rt][0][r];
            if (us_new != nullptr) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    us_new[(size_t)t * LM + L.g + 4 * r] = Uh[0][0][r];
            }
        }
        // l(x, u) = x_diff^T Q x_diff / 2 + u_diff^T R u_diff / 2   (ilqr_cost.py:532-553)
        {
            d4 xd[2][1], ud[1][1], ax[2][1], au[1][1], sx[1][1], su[1][1];
            xd[0][0] = Xh[0][0] - xg[0];
            xd[1][0] = Xh[1][0] - xg[1];
            ud[0][0] = Uh[0][0] - ug;
            mm_tn<2, 2, 1>(Qn, xd, ax);    // (x_diff^T Q)[j]
            mm_tn<2, 1, 1>(ax, xd, sx);    // its diagonal: (x_diff^T Q) . x_diff per column
            mm_tn<1, 1, 1, 2>(Rn, ud, au);
            mm_tn<1, 1, 1, 2>(au, ud, su);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (diag_lane[r])
                    J = J + (sx[0][0][r] * 0.5 + su[0][0][r] * 0.5);
        }
        // x+ = x + dt (A x + B u + c phi(W x))
        {
            d4 z[2][1], axv[2][1], bu[2][1];
            mm_tn<2, 2, 1>(WT, Xh, z);
            mm_tn<2, 2, 1>(AT, Xh, axv);
            mm_tn<1, 2, 1, 2>(BT, Uh, bu);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double zz = z[rt][0][r];
                    double s = 1.0 + zz * zz;
                    double phi = zz / sqrt(s);
                    Xh[rt][0][r] = Xh[rt][0][r] + P.dt * ((axv[rt][0][r] + bu[rt][0][r]) + P.sig_c * phi);
                }
        }
    }
    if (store_col >= 0 && L.c == store_col) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                xs_new[(size_t)T * LN + 16 * rt + L.g + 4 * r] = Xh[rt][0][r];
    }
    {
        d4 xd[2][1], ax[2][1], sx[1][1];
        xd[0][0] = Xh[0][0] - xg[0];
        xd[1][0] = Xh[1][0] - xg[1];
        mm_tn<2, 2, 1>(Qtn, xd, ax);
        mm_tn<2, 1, 1>(ax, xd, sx);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (diag_lane[r]) {
                J = J + sx[0][0][r] * 0.5;
                S.J[L.c] = J;
            }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ kernels

__global__ __launch_bounds__(64) void k_large_rollout_cost(LargeP P, int64_t B, const double* x0, const double* us,
                                                            double* xs, double* J)
{
    __shared__ WaveMem S;
    int64_t b = blockIdx.x;
    large_forward<true>(P, S, nullptr, us + b * P.T * LM, nullptr, nullptr, x0 + b * LN, 0.0, 0,
                        xs + b * (P.T + 1) * LN, nullptr);
    if (threadIdx.x == 0)
        J[b] = S.J[0];
}

__global__ __launch_bounds__(64) void k_large_backward(LargeP P, LQPOpts QO, int64_t B, const double* xs,
                                                        const double* us, const double* mu, double* k, double* K,
                                                        double* dJ, int32_t* info, int32_t* qp_result,
                                                        int32_t* free_mask)
{
    __shared__ WaveMem S;
    int64_t b = blockIdx.x;
    BwOut o;
    large_backward(P, QO, S, xs + b * (P.T + 1) * LN, us + b * P.T * LM, mu[b], k + b * P.T * LM,
                   K + b * P.T * LM * LN, o, qp_result ? qp_result + b * P.T : nullptr,
                   free_mask ? free_mask + b * P.T * LM : nullptr);
    if (threadIdx.x == 0) {
        dJ[b] = o.dJ1;
        dJ[B + b] = o.dJ2;
        info[b] = o.stopped_at;
        info[B + b] = o.qp_iter;
        info[2 * B + b] = o.qp_backtrack;
        info[3 * B + b] = o.n_clamped;
        info[4 * B + b] = o.rc;
    }
}

struct LAlphas {
    double a[32];
};

// J_new [B][n_alpha]; the trajectory of alpha[store_idx] (if >= 0) goes to xs_new / us_new.
__global__ __launch_bounds__(64) void k_large_forward_ls(LargeP P, LAlphas al, int n_alpha, int store_idx, int64_t B,
                                                          const double* xs, const double* us, const double* k,
                                                          const double* K, double* J_new, double* xs_new,
                                                          double* us_new)
{
    __shared__ WaveMem S;
    int64_t b = blockIdx.x;
    const LaneId L = lane_id();
    for (int pass = 0; pass * LA < n_alpha; ++pass) {
        int j = pass * LA + L.c;
        double a = j < n_alpha ? al.a[j < 32 ? j : 0] : 0.0;
        int sc = (store_idx >= pass * LA && store_idx < (pass + 1) * LA) ? store_idx - pass * LA : -1;
        large_forward<false>(P, S, xs + b * (P.T + 1) * LN, us + b * P.T * LM, k + b * P.T * LM,
                             K + b * P.T * LM * LN, xs + b * (P.T + 1) * LN, a, sc,
                             xs_new ? xs_new + b * (P.T + 1) * LN : nullptr, us_new ? us_new + b * P.T * LM : nullptr);
        if (threadIdx.x < LA && pass * LA + (int)threadIdx.x < n_alpha)
            J_new[b * n_alpha + pass * LA + threadIdx.x] = S.J[threadIdx.x];
        __syncthreads();
    }
}

struct LSolveBuf {
    double *xs1, *us1, *k1, *K1; // second copies (workspace); the first copies are the caller's arrays
};

// iLQR.update_u (ilqr_core.py:241-399), one problem per wavefront.
__global__ __launch_bounds__(64) void k_large_solve(LargeP P, LSolveOpts O, int64_t B, const double* x0,
                                                     const double* us_init, double* delta_io, double* k0, double* K0,
                                                     double* xs0, double* us0, LSolveBuf W, double* J_out,
                                                     double* mu_out, int32_t* iters_out, int32_t* status_out,
                                                     ilqr_trace tr)
{
    __shared__ WaveMem S;
    const int64_t b = blockIdx.x;
    const int T = P.T;
    const LaneId L = lane_id();
    const int lane = threadIdx.x;
    double* xsb[2] = {xs0 + b * (T + 1) * LN, W.xs1 + b * (T + 1) * LN};
    double* usb[2] = {us0 + b * T * LM, W.us1 + b * T * LM};
    double* kb[2] = {k0 + b * T * LM, W.k1 + b * T * LM};
    double* Kb[2] = {K0 + b * T * LM * LN, W.K1 + b * T * LM * LN};
    int cur = 0, gcur = 0;
    // us = u_seq_init.copy() ; xs = rollout ; J_opt                     (:269-271)
    for (int i = lane; i < T * LM; i += 64)
        usb[0][i] = us_init[b * T * LM + i];
    __syncthreads();
    large_forward<true>(P, S, nullptr, usb[0], nullptr, nullptr, x0 + b * LN, 0.0, 0, xsb[0], nullptr);
    double J = S.J[0];
    double mu = O.mu_from_array ? mu_out[b] : 1.0; // reset per call (:264); delta_ persists (:62)
    double delta = delta_io[b];
    int status = ILQR_RAN_OUT, iters = 0;
    double dJ1 = 0.0, dJ2 = 0.0, actual = 0.0; // carried over iterations, as in the reference
    __syncthreads();

    for (int it = 0; it < O.max_iter; ++it) {
        BwOut bw;
        // backward pass with the LinAlgError retry loop (:287-299); when it gives up at mu_max
        // the line search below still runs on the previous gains and Delta_J terms, as in the reference
        for (;;) {
            large_backward(P, O.qp, S, xsb[cur], usb[cur], mu, kb[gcur ^ 1], Kb[gcur ^ 1], bw, nullptr, nullptr);
            __syncthreads();
            if (bw.rc == 0) {
                gcur ^= 1;
                dJ1 = bw.dJ1;
                dJ2 = bw.dJ2;
                break;
            }
            delta = fmax(delta, 1.0) * O.delta0;
            mu = fmax(O.mu_min, mu * delta);
            if (O.mu_max != 0.0 && mu >= O.mu_max)
                break;
        }
        // line search (:301-313): the reference's serial first-hit scan, 16 candidates per pass
        int accepted = 0, j = 0;
        double Jnew = 0.0;
        bool ended = false;
        int nxt = cur ^ 1;
        {
            for (int pass = 0; pass * LA < O.n_alpha && !ended; ++pass) {
                int jj = pass * LA + L.c;
                double a = jj < O.n_alpha ? O.alphas[jj < 32 ? jj : 0] : 0.0;
                large_forward<false>(P, S, xsb[cur], usb[cur], kb[gcur], Kb[gcur], xsb[cur], a, pass == 0 ? 0 : -1,
                                     xsb[nxt], usb[nxt]);
                int win = -1;
                for (int c = 0; c < LA && pass * LA + c < O.n_alpha; ++c) {
                    j = pass * LA + c;
                    double alpha = O.alphas[j];
                    Jnew = S.J[c];
                    double expected = (alpha * dJ1 + alpha * alpha * dJ2) * -1;
                    if (expected <= 0.0) {
                        ended = true; // reject the whole iteration
                        break;
                    }
                    if ((J - Jnew) / expected > O.scale) {
                        actual = J - Jnew;
                        accepted = 1;
                        ended = true;
                        win = c;
                        break;
                    }
                }
                __syncthreads();
                if (!ended)
                    j = (pass + 1) * LA < O.n_alpha ? (pass + 1) * LA : O.n_alpha;
                if (accepted && !(pass == 0 && win == 0)) {
                    // the accepted candidate's states and controls (same arithmetic, same bits)
                    double aw = O.alphas[pass * LA + L.c < O.n_alpha ? pass * LA + L.c : 0];
                    large_forward<false>(P, S, xsb[cur], usb[cur], kb[gcur], Kb[gcur], xsb[cur], aw, win, xsb[nxt],
                                         usb[nxt]);
                }
            }
        }
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
        if (lane == 0) {
            int64_t to = (int64_t)it * B + b;
            if (tr.J) tr.J[to] = J;
            if (tr.accepted) tr.accepted[to] = accepted;
            if (tr.alpha_idx) tr.alpha_idx[to] = j;
            if (tr.mu) tr.mu[to] = mu;
            if (tr.delta) tr.delta[to] = delta;
            if (tr.qp_iter) tr.qp_iter[to] = bw.qp_iter;
            if (tr.qp_backtrack) tr.qp_backtrack[to] = bw.qp_backtrack;
            if (tr.n_clamped) tr.n_clamped[to] = bw.n_clamped;
            if (tr.stopped_at) tr.stopped_at[to] = bw.stopped_at;
        }
        iters = it + 1;
        if (converged)
            status = ILQR_CONVERGED;
        if (stop || converged)
            break;
    }
    __syncthreads();
    // results back into the caller's arrays
    if (cur == 1) {
        for (int i = lane; i < (T + 1) * LN; i += 64)
            xsb[0][i] = xsb[1][i];
        for (int i = lane; i < T * LM; i += 64)
            usb[0][i] = usb[1][i];
    }
    if (gcur == 1) {
        for (int i = lane; i < T * LM; i += 64)
            kb[0][i] = kb[1][i];
        for (int i = lane; i < T * LM * LN; i += 64)
            Kb[0][i] = Kb[1][i];
    }
    if (lane == 0) {
        delta_io[b] = delta;
        J_out[b] = J;
        mu_out[b] = mu;
        iters_out[b] = iters;
        status_out[b] = status;
    }
}

// ------------------------------------------------------------------ host side

int large_check(const ilqr_large_model* m)
{
    if (!m)
        return ILQR_ERR_ARG;
    if (m->n != LN || m->m != LM)
        return ILQR_ERR_UNSUPPORTED;
    if (m->T < 1 || !m->A || !m->B || !m->W || !m->Q || !m->R || !m->Q_terminal || !m->x_goal || !m->u_goal ||
        !m->lower || !m->upper)
        return ILQR_ERR_ARG;
    return ILQR_OK;
}

LargeP large_fill(const ilqr_large_model* m)
{
    LargeP P;
    P.A = m->A;
    P.B = m->B;
    P.W = m->W;
    P.Q = m->Q;
    P.R = m->R;
    P.Qt = m->Q_terminal;
    P.xg = m->x_goal;
    P.ug = m->u_goal;
    P.lower = m->lower;
    P.upper = m->upper;
    P.dt = m->dt;
    P.sig_c = m->sig_c;
    P.T = m->T;
    return P;
}

int large_qp(const ilqr_opts* o, LQPOpts& q)
{
    if (!o || o->qp_max_iter < 0 || !(o->qp_step_dec > 0.0) || !(o->qp_step_dec < 1.0) || !(o->qp_min_step > 0.0))
        return ILQR_ERR_ARG;
    q.max_iter = o->qp_max_iter;
    q.min_grad = o->qp_min_grad;
    q.min_rel = o->qp_min_rel_improve;
    q.step_dec = o->qp_step_dec;
    q.min_step = o->qp_min_step;
    q.armijo = o->qp_armijo;
    return ILQR_OK;
}

inline hipStream_t ls(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int lrc(hipError_t e) { return e == hipSuccess ? ILQR_OK : ILQR_ERR_HIP; }

} // namespace

extern "C" {

int ilqr_large_rollout_cost(const ilqr_large_model* model, int64_t B, const double* x0, const double* us, double* xs,
                            double* J, void* stream)
{
    int rc = large_check(model);
    if (rc != ILQR_OK)
        return rc;
    if (B < 1 || !x0 || !us || !xs || !J)
        return ILQR_ERR_ARG;
    hipLaunchKernelGGL(k_large_rollout_cost, dim3((unsigned)B), dim3(64), 0, ls(stream), large_fill(model), B, x0, us,
                       xs, J);
    return lrc(hipGetLastError());
}

int ilqr_large_backward(const ilqr_large_model* model, const ilqr_opts* opts, int64_t B, const double* xs,
                        const double* us, const double* mu, double* k, double* K, double* dJ, int32_t* info,
                        int32_t* qp_result, int32_t* free_mask, void* stream)
{
    int rc = large_check(model);
    if (rc != ILQR_OK)
        return rc;
    LQPOpts q;
    rc = large_qp(opts, q);
    if (rc != ILQR_OK)
        return rc;
    if (B < 1 || !xs || !us || !mu || !k || !K || !dJ || !info)
        return ILQR_ERR_ARG;
    hipLaunchKernelGGL(k_large_backward, dim3((unsigned)B), dim3(64), 0, ls(stream), large_fill(model), q, B, xs, us,
                       mu, k, K, dJ, info, qp_result, free_mask);
    return lrc(hipGetLastError());
}

int ilqr_large_forward_ls(const ilqr_large_model* model, int64_t B, int32_t n_alpha, const double* alphas,
                          int32_t store_idx, const double* xs, const double* us, const double* k, const double* K,
                          double* J_new, double* xs_new, double* us_new, void* stream)
{
    int rc = large_check(model);
    if (rc != ILQR_OK)
        return rc;
    if (B < 1 || n_alpha < 1 || n_alpha > 32 || !alphas || !xs || !us || !k || !K || !J_new)
        return ILQR_ERR_ARG;
    if (store_idx >= n_alpha || (store_idx >= 0 && (!xs_new || !us_new)))
        return ILQR_ERR_ARG;
    LAlphas al;
    for (int i = 0; i < 32; ++i)
        al.a[i] = i < n_alpha ? alphas[i] : 0.0;
    hipLaunchKernelGGL(k_large_forward_ls, dim3((unsigned)B), dim3(64), 0, ls(stream), large_fill(model), al,
                       (int)n_alpha, (int)store_idx, B, xs, us, k, K, J_new, xs_new, us_new);
    return lrc(hipGetLastError());
}

size_t ilqr_large_workspace_bytes(const ilqr_large_model* model, int64_t B)
{
    if (large_check(model) != ILQR_OK || B < 1)
        return 0;
    size_t T = (size_t)model->T;
    return (size_t)B * sizeof(double) * ((T + 1) * LN + T * LM + T * LM + T * LM * LN);
}

int ilqr_large_solve_batched(const ilqr_large_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                             const double* us_init, double* delta, double* k, double* K, double* xs, double* us,
                             double* J, double* mu, int32_t* iters, int32_t* status, const ilqr_trace* trace,
                             void* workspace, size_t workspace_bytes, void* stream)
{
    int rc = large_check(model);
    if (rc != ILQR_OK)
        return rc;
    if (!opts || !opts->alphas || opts->n_alpha < 0 || opts->n_alpha > 32 || opts->max_iter < 0)
        return ILQR_ERR_ARG;
    if (B < 1 || !x0 || !us_init || !delta || !k || !K || !xs || !us || !J || !mu || !iters || !status || !workspace)
        return ILQR_ERR_ARG;
    if (workspace_bytes < ilqr_large_workspace_bytes(model, B))
        return ILQR_ERR_WORKSPACE;
    LSolveOpts O;
    O.max_iter = opts->max_iter;
    O.n_alpha = opts->n_alpha;
    O.mu_from_array = (opts->mu_init >= 0.0) ? 1 : 0;
    O.tol = opts->tol;
    O.scale = opts->scale;
    O.mu_min = opts->mu_min;
    O.mu_max = opts->mu_max;
    O.delta0 = opts->delta0;
    for (int i = 0; i < 32; ++i)
        O.alphas[i] = i < opts->n_alpha ? opts->alphas[i] : 0.0;
    rc = large_qp(opts, O.qp);
    if (rc != ILQR_OK)
        return rc;
    size_t T = (size_t)model->T;
    LSolveBuf W;
    double* p = static_cast<double*>(workspace);
    W.xs1 = p;
    p += (size_t)B * (T + 1) * LN;
    W.us1 = p;
    p += (size_t)B * T * LM;
    W.k1 = p;
    p += (size_t)B * T * LM;
    W.K1 = p;
    ilqr_trace tr;
    if (trace)
        tr = *trace;
    else
        std::memset(&tr, 0, sizeof(tr));
    hipLaunchKernelGGL(k_large_solve, dim3((unsigned)B), dim3(64), 0, ls(stream), large_fill(model), O, B, x0, us_init,
                       delta, k, K, xs, us, W, J, mu, iters, status, tr);
    return lrc(hipGetLastError());
}

} // extern "C"
