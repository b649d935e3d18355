// Wide path of the batched control-limited iLQR solve: up to 64 states and 16 controls, one PROBLEM PER
// WORKGROUP (four wavefronts), every matrix of a backward step resident in the workgroup's LDS.
//
// The reference's sweep is dimension-agnostic (ilqr/ilqr_core.py:109-228); the register-resident path of
// ilqr_large.hip stops at 32 x 8 because a wavefront's registers hold no more than four 32 x 32 matrices.  Here
// V_xx, f_x and M1 = f_x^T V_xx live in LDS as row-major arrays, the matrix products are fp64 MFMAs
// (v_mfma_f64_16x16x4_f64) whose operands are read from LDS -- one double per lane and k-block -- and whose
// 16 x 16 output tiles are dealt round-robin to the four wavefronts; what is a chain per row (matrix-vector
// products, the cost gradients, the dynamics) runs on one lane per row; boxQP runs on one wavefront with one
// lane per variable and, in its Armijo loop, one lane per trip.
//
// Arithmetic contract (the same as the other paths): v_mfma_f64_16x16x4_f64 is the k-ascending chain
// fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C)))), so a product accumulated over consecutive k-blocks
// from C = 0 is the "first product, then fma" chain of the CPU restatement of the reference's np.dot calls
// (only the sign of an exact zero can differ); operands outside the n x n / m x n / m x m matrices read as
// exact zeros, which append fma(0, 0, s) == s to the end of a chain.  Everything else is plain unfused fp64 in
// the reference's order (compile with -ffp-contract=off).
//
// MFMA operand layouts (lane l, g = l >> 4, c = l & 15): the A operand of k-block kb is A[16 it + c][4 kb + g],
// the B operand is B[4 kb + g][16 jt + c], register r of the result is D[16 it + 4 r + g][16 jt + c].
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/ilqr_hip.h"

namespace {

constexpr int XN = 64;  // most states
constexpr int XM = 16;  // most controls
constexpr int LDN = 65; // row stride (doubles) of a matrix with n columns in LDS; odd, so that a column is read
constexpr int LDM = 17; // without bank conflicts as well as a row; same for m columns

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 xl_mfma(double a, double b, d4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

struct XlP { // kernel arguments: the model, all device pointers, arrays dense ([n*n], [n*m], ...)
    const double *A, *B, *W, *Q, *R, *Qt, *xg, *ug, *lower, *upper;
    double dt, sig_c;
    int n, m, T;
    int path;         // PathQRCost: xg [T+1][n], ug [T][m] hold the goal of every time step, values carry no 1/2
    int64_t gsx, gsu; // doubles between the goals (or paths) of consecutive problems; 0: shared by the batch
};

__device__ __forceinline__ XlP xl_for_problem(const XlP& P, int64_t b)
{
    XlP Q = P;
    Q.xg = P.xg + b * P.gsx;
    Q.ug = P.ug + b * P.gsu;
    return Q;
}

struct XlQP {
    int max_iter;
    double min_grad, min_rel, step_dec, min_step, armijo;
};

struct XlSolveOpts {
    int max_iter, n_alpha, mu_from_array;
    double tol, scale, mu_min, mu_max, delta0;
    double alphas[32];
    XlQP qp;
    double* dJ_carry;
};

// ---- LDS of one workgroup = one problem ----------------------------------------------------------------------

struct XlBw { // a backward sweep
    double Vxx[XN * LDN], fx[XN * LDN];
    double M1[XN * LDN]; // f_x^T V_xx; later in the step the unsymmetrised V_xx
    double fu[XN * LDM];
    union {
        double M2[XM * LDN]; // f_u^T (V_xx + mu I)
        double T1[XN * LDM]; // K^T Q_uu (after Q_ux and Q_uu have been formed from M2)
    };
    double Qux[XM * LDN], Kl[XM * LDN];
    double H[XM * LDM];  // Q_uu
    double Lf[XM * LDM]; // Cholesky factor: of Q_uu for the warm start, then boxQP's H_ff (rows/columns of the free set)
    double x[XN], xd[XN], lx[XN], gi[XN], Vx[XN], Qx[XN];
    double u[XM], ud[XM], lu[XM], Qu[XM], lo[XM], hi[XM], qx[XM], grad[XM], srch[XM], rd[XM], qa[XM];
};

struct XlFwWave { // a rollout of one wavefront
    double xn[XN], dv[XN], acc[XN], un[XM], ud[XM];
};

struct XlFw { // the rollouts of a line search: the model transposed (a lane per row reads consecutive words)
    double At[XN * LDN], Wt[XN * LDN], Qr[XN * LDN], Bt[XM * LDN];
    XlFwWave wv[4];
};

struct XlMem {
    union {
        XlBw bw;
        XlFw fw;
    };
    double Jw[4];      // cost of each wavefront's rollout
    double res_d[4];   // wave 0 -> workgroup: value hand-overs (boxQP: candidate value; sweep: Delta_J1, Delta_J2)
    int res_i[8];      // ... and integers (flag of a backward step; counters of a sweep)
    int next;          // the problem taken from the queue
};

static_assert(sizeof(XlMem) <= 160 * 1024, "one problem's matrices must fit the LDS of a CU");

__device__ __forceinline__ void xl_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double xl_clamp(double lo, double hi, double v)
{
    // np.maximum(lower, np.minimum(upper, x)) with NaN propagation (ilqr_boxQP.py:37)
    double t = (v != v) ? v : (hi < v ? hi : v);
    return (t != t) ? t : (lo > t ? lo : t);
}

// ---- matrix products out of LDS ----------------------------------------------------------------------------------

struct XlMat { // logical element (r, c): p[r * ld + c], or p[c * ld + r] when tr; zero outside rows x cols
    const double* p;
    int ld, rows, cols;
    bool tr;
};

__device__ __forceinline__ double xl_at(const XlMat& X, int r, int c)
{
    return (r < X.rows && c < X.cols) ? (X.tr ? X.p[c * X.ld + r] : X.p[r * X.ld + c]) : 0.0;
}

// Tile (it, jt) of A B', A: I x kd, B' = B + diag I: kd x J; the chain of every element runs over k = 0 ... kd-1.
template <bool DIAG = false>
__device__ __forceinline__ d4 xl_tile(const XlMat& A, const XlMat& Bm, int it, int jt, int kd, int g, int c,
                                      double diag = 0.0)
{
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    const int i = 16 * it + c, j = 16 * jt + c;
    for (int k0 = 0; k0 < kd; k0 += 4) {
        const int k = k0 + g;
        const double a = xl_at(A, i, k);
        double b = xl_at(Bm, k, j);
        if (DIAG && k < kd)
            b = b + (k == j ? diag : 0.0); // V_xx + mu * eye   (ilqr_core.py:210)
        acc = xl_mfma(a, b, acc);
    }
    return acc;
}

__device__ __forceinline__ void xl_tile_store(double* O, int ld, int rows, int cols, int it, int jt, d4 v, int g, int c)
{
    const int col = 16 * jt + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * it + 4 * r + g;
        if (row < rows && col < cols)
            O[row * ld + col] = v[r];
    }
}

// ---- Cholesky and triangular solves of the free block (m <= 16) ------------------------------------------------

// Lower Cholesky factor of the rows/columns `fm` of H (what the reference factorises after compacting the free
// variables, ilqr_boxQP.py:114-116: the same chains, the clamped indices skipped), run by one wavefront: lane a owns
// row a; column by column, the pivot's square root and reciprocal, then every row below it in parallel -- each
// entry the k-ascending chain t = fma(-L[a][k], L[j][k], t) of LAPACK's dpotf2, scaled by the reciprocal.
// Reads the lower triangle of H.  False on a non-positive (or NaN) pivot.
__device__ bool xl_chol(const double* H, double* Lf, double* rd, int m, unsigned fm, int lane)
{
    for (int j = 0; j < m; ++j) {
        if (!((fm >> j) & 1u))
            continue;
        const bool mine = lane < m && lane >= j && ((fm >> lane) & 1u);
        double t = 0.0;
        if (mine) {
            t = H[lane * LDM + j];
            for (int k = 0; k < j; ++k)
                if ((fm >> k) & 1u)
                    t = fma(-Lf[lane * LDM + k], Lf[j * LDM + k], t);
        }
        const double s = __shfl(t, j);
        if (!(s > 0.0))
            return false;
        const double d = sqrt(s);
        const double r = 1.0 / d;
        if (mine)
            Lf[lane * LDM + j] = (lane == j) ? d : t * r;
        if (lane == j)
            rd[j] = r;
        xl_wave_sync();
    }
    return true;
}

// (L L^T) x = b in place for the rows/columns `fm` (cho_solve, each substitution multiplying by the reciprocal of
// the diagonal entry); every lane solves its own right-hand side, the factor is read from LDS.
__device__ __forceinline__ void xl_chol_solve(const double* Lf, const double* rd, int m, unsigned fm, double (&b)[XM])
{
#pragma unroll
    for (int i = 0; i < XM; ++i) {
        if (i < m && ((fm >> i) & 1u)) {
            double s = b[i];
#pragma unroll
            for (int j = 0; j < i; ++j)
                if ((fm >> j) & 1u)
                    s = fma(-Lf[i * LDM + j], b[j], s);
            b[i] = s * rd[i];
        }
    }
#pragma unroll
    for (int i = XM - 1; i >= 0; --i) {
        if (i < m && ((fm >> i) & 1u)) {
            double s = b[i];
#pragma unroll
            for (int j = i + 1; j < XM; ++j)
                if (j < m && ((fm >> j) & 1u))
                    s = fma(-Lf[j * LDM + i], b[j], s);
            b[i] = s * rd[i];
        }
    }
}

// ---- boxQP (ilqr_boxQP.py:9-186) on one wavefront -----------------------------------------------------------

struct XlQpRes {
    int result;
    unsigned fm; // free variables at the last classification
    int n_iter, n_factor, n_backtrack;
    bool ok; // false: the reference would have raised LinAlgError
};

// objective x.g + 0.5 x^T H x of a candidate held in registers (:62, :148): every chain in the reference's order
__device__ __forceinline__ double xl_value(const XlBw& W, int m, const double (&x)[XM])
{
    double xg = x[0] * W.Qu[0];
#pragma unroll
    for (int i = 1; i < XM; ++i)
        if (i < m)
            xg = fma(x[i], W.Qu[i], xg);
    double tot = 0.0;
#pragma unroll
    for (int j = 0; j < XM; ++j) {
        if (j < m) {
            double acc = x[0] * W.H[0 * LDM + j];
#pragma unroll
            for (int i = 1; i < XM; ++i)
                if (i < m)
                    acc = fma(x[i], W.H[i * LDM + j], acc);
            tot = (j == 0) ? acc * x[0] : fma(acc, x[j], tot);
        }
    }
    return xg + 0.5 * tot;
}

// In: W.H (Q_uu, finite and positive definite: the caller has factorised it), W.Qu (g), W.lo, W.hi, W.qx (x0).
// Out: W.qx (x), W.Lf / W.rd (H_ff of the free set; all ones if no factorisation happened, :36), the result.
// Lane i < m owns variable i for the gradient and the classification; the Armijo loop (:146-156) evaluates 64
// consecutive trips at a time, one per lane, and takes the first that ends the loop.
__device__ XlQpRes xl_boxqp(XlBw& W, const XlQP& O, int m, int lane, double* hand)
{
    const unsigned all = (m >= 32) ? 0xffffffffu : ((1u << m) - 1u);
    XlQpRes R{-1, 0u, 0, 0, 0, true};
    unsigned cm = 0u;
    for (int e = lane; e < XM * LDM; e += 64)
        W.Lf[e] = 1.0;
    if (lane < XM)
        W.rd[lane] = 1.0;
    double lo = 0.0, hi = 0.0, gl = 0.0;
    if (lane < m) {
        lo = W.lo[lane];
        hi = W.hi[lane];
        gl = W.Qu[lane];
        W.qx[lane] = xl_clamp(lo, hi, W.qx[lane]);
    }
    xl_wave_sync();
    double value;
    {
        double x0r[XM];
#pragma unroll
        for (int i = 0; i < XM; ++i)
            x0r[i] = i < m ? W.qx[i] : 0.0;
        value = xl_value(W, m, x0r);
    }
    double oldvalue = 0.0;
    for (int iter = 0; iter < O.max_iter; ++iter) {
        if (iter > 1 && (oldvalue - value) < O.min_rel * fabs(oldvalue) && (oldvalue - value) > 0) {
            R.result = 4;
            break;
        }
        oldvalue = value;
        // grad = g + H x (:82); clamped = at a bound with the gradient pointing outwards (:85-88)
        bool cl = false;
        if (lane < m) {
            const double xi = W.qx[lane];
            double acc = W.H[lane * LDM + 0] * W.qx[0];
            for (int j = 1; j < m; ++j)
                acc = fma(W.H[lane * LDM + j], W.qx[j], acc);
            const double grad = gl + acc;
            W.grad[lane] = grad;
            cl = ((xi == lo) && (grad > 0)) || ((xi == hi) && (grad < 0));
        }
        const unsigned oldcm = cm;
        cm = (unsigned)__ballot(cl) & all;
        const unsigned fm = all & ~cm;
        R.fm = fm;
        xl_wave_sync();
        if (cm == all) {
            R.result = 6;
            break;
        }
        if (iter == 1 || oldcm != cm) { // :107-116
            if (!xl_chol(W.H, W.Lf, W.rd, m, fm, lane)) {
                R.ok = false;
                return R;
            }
            R.n_factor++;
        }
        // sla.norm(grad[free]) (:120); dnrm2 of a single element is its magnitude
        double ss = 0.0, gfirst = 0.0;
        bool first = true;
        for (int a = 0; a < m; ++a)
            if ((fm >> a) & 1u) {
                const double ga = W.grad[a];
                if (first) {
                    ss = ga * ga;
                    gfirst = ga;
                    first = false;
                } else {
                    ss = fma(ga, ga, ss);
                }
            }
        const double gnorm = __popc(fm) == 1 ? fabs(gfirst) : sqrt(ss);
        if (gnorm < O.min_grad) {
            R.result = 5;
            break;
        }
        R.n_iter++;
        // search[free] = -cho_solve(H_ff, g[free] + H_fc x[clamped]) - x[free]   (:126-132)
        double b[XM];
#pragma unroll
        for (int a = 0; a < XM; ++a) {
            b[a] = 0.0;
            if (a < m && ((fm >> a) & 1u)) {
                double acc = 0.0;
                bool fst = true;
                for (int q = 0; q < m; ++q)
                    if ((cm >> q) & 1u) {
                        acc = fst ? W.H[a * LDM + q] * W.qx[q] : fma(W.H[a * LDM + q], W.qx[q], acc);
                        fst = false;
                    }
                b[a] = W.Qu[a] + acc;
            }
        }
        xl_chol_solve(W.Lf, W.rd, m, fm, b);
#pragma unroll
        for (int a = 0; a < XM; ++a)
            if (a < m)
                W.srch[a] = ((fm >> a) & 1u) ? b[a] * -1 - W.qx[a] : 0.0; // (every lane writes the same number)
        xl_wave_sync();
        // Armijo loop (:141-156): trip t tries step_dec^t (formed by repeated multiplication, as the loop does);
        // it ends at the first trip whose test fails or, from the second trip on, whose step is below min_step.
        double base = 1.0;
        int t0 = 0;
        for (;;) {
            double st = base, mine = base;
            for (int q = 0; q < 64; ++q) {
                if (q == lane)
                    mine = st;
                st = st * O.step_dec;
            }
            double xc[XM];
#pragma unroll
            for (int a = 0; a < XM; ++a)
                xc[a] = a < m ? xl_clamp(W.lo[a], W.hi[a], W.qx[a] + mine * W.srch[a]) : 0.0;
            double sd = (xc[0] - W.qx[0]) * W.grad[0];
#pragma unroll
            for (int a = 1; a < XM; ++a)
                if (a < m)
                    sd = sd + (xc[a] - W.qx[a]) * W.grad[a];
            const double vc = xl_value(W, m, xc);
            const bool cont = (vc - oldvalue) / (sd + 1e-10) < O.armijo;
            const bool exh = (t0 + lane >= 1) && (mine < O.min_step);
            const unsigned long long bal = __ballot(!cont || exh);
            if (bal != 0ull) {
                const int l = __ffsll((long long)bal) - 1;
                xl_wave_sync(); // every lane has read the old x
                if (lane == l) {
#pragma unroll
                    for (int a = 0; a < XM; ++a)
                        if (a < m)
                            W.qx[a] = xc[a];
                    hand[0] = vc;
                    hand[1] = exh ? 2.0 : 0.0;
                }
                xl_wave_sync();
                value = hand[0];
                if (hand[1] != 0.0)
                    R.result = 2;
                R.n_backtrack += t0 + l;
                xl_wave_sync();
                break;
            }
            t0 += 64;
            base = st;
        }
        if (R.result != -1)
            break;
    }
    return R;
}

// ---- backward sweep ---------------------------------------------------------------------------------------

struct XlBwOut {
    double dJ1, dJ2;
    int stopped_at, qp_iter, qp_backtrack, n_clamped, rc;
};

enum { XL_STEP_OK = 0, XL_STEP_NOT_PD = 1, XL_STEP_LINALG = 2 };

// _backward_pass + _compute_Q_terms (ilqr_core.py:109-228) of one problem, run by the whole workgroup.
// k [T*m], K [T*m*n] are fully written (zeros where the sweep did not reach).
template <bool PATH>
__device__ void xl_backward(const XlP& P, const XlQP& QO, XlMem& S, const double* __restrict__ xs,
                            const double* __restrict__ us, double mu, double* __restrict__ k, double* __restrict__ K,
                            XlBwOut& o, int32_t* qp_result, int32_t* free_mask)
{
    XlBw& W = S.bw;
    const int n = P.n, m = P.m, T = P.T, NT = (n + 15) >> 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
    if (qp_result)
        for (int i = tid; i < T; i += 256)
            qp_result[i] = -100; // step not reached
    if (free_mask)
        for (int i = tid; i < T * m; i += 256)
            free_mask[i] = 0;
    // V_x = l_x(x_N), V_xx = l_xx(x_N)   (:127-128)
    if (tid < n)
        W.xd[tid] = xs[(int64_t)T * n + tid] - (PATH ? P.xg[(int64_t)T * n + tid] : P.xg[tid]);
    __syncthreads();
    if (tid < n) {
        const int j = tid;
        double acc = W.xd[0] * (P.Qt[0 * n + j] + P.Qt[j * n + 0]);
        for (int a = 1; a < n; ++a)
            acc = fma(W.xd[a], P.Qt[a * n + j] + P.Qt[j * n + a], acc);
        W.Vx[j] = acc;
    }
    for (int e = tid; e < n * n; e += 256) {
        const int a = e / n, b = e % n;
        W.Vxx[a * LDN + b] = P.Qt[a * n + b] + P.Qt[b * n + a];
    }
    double dJ1 = 0.0, dJ2 = 0.0; // (wave 0's; handed to the workgroup at the end)
    int qp_iter = 0, qp_bt = 0, n_cl = 0;
    o.stopped_at = -1;
    o.rc = 0;
    __syncthreads();

    for (int i = T - 1; i >= 0; --i) {
        const double* xg = PATH ? P.xg + (int64_t)i * n : P.xg;
        const double* ug = PATH ? P.ug + (int64_t)i * m : P.ug;
        // -- operands of the step
        if (tid < n) {
            const double xv = xs[(int64_t)i * n + tid];
            W.x[tid] = xv;
            W.xd[tid] = xv - xg[tid];
        } else if (tid >= 64 && tid < 64 + m) {
            const int j = tid - 64;
            const double uv = us[(int64_t)i * m + j];
            W.u[j] = uv;
            W.ud[j] = uv - ug[j];
            W.lo[j] = P.lower[j] - uv;
            W.hi[j] = P.upper[j] - uv;
        }
        if (tid == 0)
            S.res_i[0] = XL_STEP_OK;
        __syncthreads();
        // -- chains per row: phi'(W x), l_x, l_u
        if (wave == 0) {
            if (lane < n) {
                double z = P.W[lane * n + 0] * W.x[0];
                for (int j = 1; j < n; ++j)
                    z = fma(P.W[lane * n + j], W.x[j], z);
                const double s = 1.0 + z * z;
                W.gi[lane] = P.sig_c * (1.0 / (s * sqrt(s)));
            }
        } else if (wave == 1) {
            if (lane < n) {
                const int j = lane;
                double acc = W.xd[0] * (P.Q[0 * n + j] + P.Q[j * n + 0]);
                for (int a = 1; a < n; ++a)
                    acc = fma(W.xd[a], P.Q[a * n + j] + P.Q[j * n + a], acc);
                W.lx[j] = acc;
            }
        } else if (wave == 2) {
            if (lane < m) {
                const int j = lane;
                double acc = W.ud[0] * (P.R[0 * m + j] + P.R[j * m + 0]);
                for (int a = 1; a < m; ++a)
                    acc = fma(W.ud[a], P.R[a * m + j] + P.R[j * m + a], acc);
                W.lu[j] = acc;
            }
        }
        __syncthreads();
        // -- f_x = I + dt (A + c diag(phi'(W x)) W), f_u = dt B
        for (int e = tid; e < n * n; e += 256) {
            const int a = e / n, b = e % n;
            W.fx[a * LDN + b] = (a == b ? 1.0 : 0.0) + P.dt * (P.A[e] + W.gi[a] * P.W[e]);
        }
        for (int e = tid; e < n * m; e += 256)
            W.fu[(e / m) * LDM + e % m] = P.dt * P.B[e];
        __syncthreads();
        // -- M1 = f_x^T V_xx, M2 = f_u^T (V_xx + mu I); Q_x, Q_u   (:200-212)
        {
            const XlMat fxT{W.fx, LDN, n, n, true}, fuT{W.fu, LDM, m, n, true}, Vm{W.Vxx, LDN, n, n, false};
            for (int q = wave; q < NT * NT + NT; q += 4) {
                if (q < NT * NT) {
                    const int it = q / NT, jt = q % NT;
                    xl_tile_store(W.M1, LDN, n, n, it, jt, xl_tile(fxT, Vm, it, jt, n, g, c), g, c);
                } else {
                    const int jt = q - NT * NT;
                    xl_tile_store(W.M2, LDN, m, n, 0, jt, xl_tile<true>(fuT, Vm, 0, jt, n, g, c, mu), g, c);
                }
            }
            if (wave == 3) {
                if (lane < n) {
                    double acc = W.fx[0 * LDN + lane] * W.Vx[0];
                    for (int a = 1; a < n; ++a)
                        acc = fma(W.fx[a * LDN + lane], W.Vx[a], acc);
                    W.Qx[lane] = W.lx[lane] + acc;
                }
            } else if (wave == 2) {
                if (lane < m) {
                    double acc = W.fu[0 * LDM + lane] * W.Vx[0];
                    for (int a = 1; a < n; ++a)
                        acc = fma(W.fu[a * LDM + lane], W.Vx[a], acc);
                    W.Qu[lane] = W.lu[lane] + acc;
                }
            }
        }
        __syncthreads();
        // -- Q_xx = l_xx + M1 f_x (kept in registers by the wavefront that owns the tile), Q_ux = M2 f_x,
        //    Q_uu = l_uu + M2 f_u
        d4 qxx[4];
        {
            const XlMat M1m{W.M1, LDN, n, n, false}, M2m{W.M2, LDN, m, n, false}, fxm{W.fx, LDN, n, n, false},
                fum{W.fu, LDM, n, m, false};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int q = wave + 4 * s;
                qxx[s] = d4{0.0, 0.0, 0.0, 0.0};
                if (q < NT * NT) {
                    const int it = q / NT, jt = q % NT;
                    const d4 acc = xl_tile(M1m, fxm, it, jt, n, g, c);
                    const int b = 16 * jt + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a = 16 * it + 4 * r + g;
                        qxx[s][r] = (a < n && b < n) ? (P.Q[a * n + b] + P.Q[b * n + a]) + acc[r] : 0.0;
                    }
                }
            }
            for (int jt = wave; jt < NT; jt += 4) {
                d4 acc = xl_tile(M2m, fxm, 0, jt, n, g, c);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = 0.0 + acc[r]; // l_ux is identically zero
                xl_tile_store(W.Qux, LDN, m, n, 0, jt, acc, g, c);
            }
            if (wave == (NT < 4 ? NT : 3)) { // (the wavefront with the fewest tiles above)
                d4 acc = xl_tile(M2m, fum, 0, 0, n, g, c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = 4 * r + g;
                    if (a < m && c < m)
                        acc[r] = (P.R[a * m + c] + P.R[c * m + a]) + acc[r];
                }
                xl_tile_store(W.H, LDM, m, m, 0, 0, acc, g, c);
            }
        }
        __syncthreads();
        // -- wave 0: PD test, warm start, boxQP, gains, Delta_J   (:134-176)
        if (wave == 0) {
            bool bad = false;
            for (int e = lane; e < m * m; e += 64)
                bad = bad || !(fabs(W.H[(e / m) * LDM + e % m]) < __builtin_inf());
            const unsigned all = (1u << m) - 1u;
            int flag = XL_STEP_OK;
            if (__any(bad)) {
                flag = XL_STEP_LINALG; // eigvals raises LinAlgError on NaN / Inf
            } else if (!xl_chol(W.H, W.Lf, W.rd, m, all, lane)) {
                flag = XL_STEP_NOT_PD;
            } else {
                // k = cho_solve(R_Q_uu, Q_u): the warm start of boxQP (positive sign, as the reference has it)
                double b[XM];
#pragma unroll
                for (int a = 0; a < XM; ++a)
                    b[a] = a < m ? W.Qu[a] : 0.0;
                xl_chol_solve(W.Lf, W.rd, m, all, b);
#pragma unroll
                for (int a = 0; a < XM; ++a)
                    if (a < m)
                        W.qx[a] = b[a];
                xl_wave_sync();
                const XlQpRes R = xl_boxqp(W, QO, m, lane, S.res_d);
                if (!R.ok) {
                    flag = XL_STEP_LINALG;
                } else {
                    qp_iter += R.n_iter;
                    qp_bt += R.n_backtrack;
                    if (__popc(R.fm) < m)
                        ++n_cl;
                    if (lane < m) {
                        k[(int64_t)i * m + lane] = W.qx[lane];
                        if (free_mask)
                            free_mask[(int64_t)i * m + lane] = (int32_t)((R.fm >> lane) & 1u);
                    }
                    if (lane == 0 && qp_result)
                        qp_result[i] = R.result;
                    // K[free, :] = -cho_solve(H_ff, Q_ux[free, :]), zero rows for the clamped controls (:157-160)
                    if (lane < n) {
                        double col[XM];
#pragma unroll
                        for (int a = 0; a < XM; ++a)
                            col[a] = (a < m && ((R.fm >> a) & 1u)) ? W.Qux[a * LDN + lane] : 0.0;
                        if (R.fm != 0u)
                            xl_chol_solve(W.Lf, W.rd, m, R.fm, col);
#pragma unroll
                        for (int a = 0; a < XM; ++a)
                            if (a < m) {
                                const double v = ((R.fm >> a) & 1u) ? col[a] * -1 : 0.0;
                                W.Kl[a * LDN + lane] = v;
                                K[((int64_t)i * m + a) * n + lane] = v;
                            }
                    }
                    // Delta_J1 += k.Q_u ; Delta_J2 += 0.5 k.(Q_uu k)   (:175-176)
                    if (lane < m) {
                        double acc = W.H[lane * LDM + 0] * W.qx[0];
                        for (int q = 1; q < m; ++q)
                            acc = fma(W.H[lane * LDM + q], W.qx[q], acc);
                        W.qa[lane] = acc;
                    }
                    xl_wave_sync();
                    double d1 = W.qx[0] * W.Qu[0];
                    for (int q = 1; q < m; ++q)
                        d1 = fma(W.qx[q], W.Qu[q], d1);
                    double d2 = W.qx[0] * W.qa[0];
                    for (int q = 1; q < m; ++q)
                        d2 = fma(W.qx[q], W.qa[q], d2);
                    dJ1 = dJ1 + d1;
                    dJ2 = dJ2 + d2 * 0.5;
                }
            }
            if (lane == 0)
                S.res_i[0] = flag;
        }
        __syncthreads();
        const int flag = S.res_i[0];
        if (flag == XL_STEP_LINALG) {
            o.rc = -1;
            break;
        }
        if (flag == XL_STEP_NOT_PD) { // print / warn / break (:137-140): the steps not reached keep zero gains
            o.stopped_at = i;
            for (int64_t e = tid; e < (int64_t)(i + 1) * m; e += 256)
                k[e] = 0.0;
            for (int64_t e = tid; e < (int64_t)(i + 1) * m * n; e += 256)
                K[e] = 0.0;
            break;
        }
        // -- T1 = K^T Q_uu
        {
            const XlMat KT{W.Kl, LDN, n, m, true}, Hm{W.H, LDM, m, m, false};
            for (int it = wave; it < NT; it += 4)
                xl_tile_store(W.T1, LDM, n, m, it, 0, xl_tile(KT, Hm, it, 0, m, g, c), g, c);
        }
        __syncthreads();
        // -- V_xx = Q_xx + (K^T Q_uu) K + K^T Q_ux + Q_ux^T K (unsymmetrised, into M1's place); V_x   (:164-167)
        {
            const XlMat T1m{W.T1, LDM, n, m, false}, Km{W.Kl, LDN, m, n, false}, KT{W.Kl, LDN, n, m, true},
                Quxm{W.Qux, LDN, m, n, false}, QuxT{W.Qux, LDN, n, m, true};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int q = wave + 4 * s;
                if (q < NT * NT) {
                    const int it = q / NT, jt = q % NT;
                    const d4 t1 = xl_tile(T1m, Km, it, jt, m, g, c);
                    const d4 t2 = xl_tile(KT, Quxm, it, jt, m, g, c);
                    const d4 t3 = xl_tile(QuxT, Km, it, jt, m, g, c);
                    d4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = ((qxx[s][r] + t1[r]) + t2[r]) + t3[r];
                    xl_tile_store(W.M1, LDN, n, n, it, jt, v, g, c);
                }
            }
            if (wave == 0 && lane < n) {
                const int a = lane;
                double t1 = W.T1[a * LDM + 0] * W.qx[0];
                double t2 = W.Kl[0 * LDN + a] * W.Qu[0];
                double t3 = W.Qux[0 * LDN + a] * W.qx[0];
                for (int q = 1; q < m; ++q) {
                    t1 = fma(W.T1[a * LDM + q], W.qx[q], t1);
                    t2 = fma(W.Kl[q * LDN + a], W.Qu[q], t2);
                    t3 = fma(W.Qux[q * LDN + a], W.qx[q], t3);
                }
                W.Vx[a] = ((W.Qx[a] + t1) + t2) + t3;
            }
        }
        __syncthreads();
        // -- V_xx = 0.5 (V_xx + V_xx^T)   (:168)
        for (int e = tid; e < n * n; e += 256) {
            const int a = e / n, b = e % n;
            W.Vxx[a * LDN + b] = 0.5 * (W.M1[a * LDN + b] + W.M1[b * LDN + a]);
        }
        __syncthreads();
    }
    // wave 0's sums to everybody
    __syncthreads();
    if (tid == 0) {
        S.res_d[2] = dJ1;
        S.res_d[3] = dJ2;
        S.res_i[1] = qp_iter;
        S.res_i[2] = qp_bt;
        S.res_i[3] = n_cl;
    }
    __syncthreads();
    o.dJ1 = S.res_d[2];
    o.dJ2 = S.res_d[3];
    o.qp_iter = S.res_i[1];
    o.qp_backtrack = S.res_i[2];
    o.n_clamped = S.res_i[3];
    __syncthreads();
}

// ---- rollouts ---------------------------------------------------------------------------------------------

// The model in the form the rollouts read it (all threads of the workgroup; ends with a barrier).
__device__ void xl_forward_load(const XlP& P, XlMem& S)
{
    XlFw& F = S.fw;
    const int n = P.n, m = P.m;
    __syncthreads();
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e / n, j = e % n;
        F.At[j * LDN + i] = P.A[e];
        F.Wt[j * LDN + i] = P.W[e];
        F.Qr[i * LDN + j] = P.Q[e];
    }
    for (int e = threadIdx.x; e < n * m; e += 256)
        F.Bt[(e % m) * LDN + e / m] = P.B[e];
    __syncthreads();
}

// v^T M v as (v^T M) . v (the order of x_diff.T.dot(Q).dot(x_diff)): lane j forms column j's chain, then every lane
// the chain over the columns.  v (len entries) and acc are the wavefront's LDS arrays; M [len][ld].
__device__ __forceinline__ double xl_quad(int len, const double* v, const double* M, int ld, double* acc, int lane)
{
    if (lane < len) {
        double a = v[0] * M[0 * ld + lane];
        for (int i = 1; i < len; ++i)
            a = fma(v[i], M[i * ld + lane], a);
        acc[lane] = a;
    }
    xl_wave_sync();
    double tot = acc[0] * v[0];
    for (int j = 1; j < len; ++j)
        tot = fma(acc[j], v[j], tot);
    xl_wave_sync();
    return tot;
}

// One rollout by one wavefront.  k == nullptr: _dynamics_rollout + _trajectory_cost (ilqr_core.py:69-83, :230-239)
// of the controls `us`; else _forward_pass (:85-107) with step size alpha around (xs, us), no clamp on the new
// controls, plus the cost.  xs_new / us_new may be null (nothing stored).  Returns J (the same number on every lane).
template <bool PATH>
__device__ double xl_forward(const XlP& P, XlMem& S, const double* __restrict__ xs, const double* __restrict__ us,
                             const double* __restrict__ k, const double* __restrict__ K, const double* __restrict__ x0,
                             double alpha, double* __restrict__ xs_new, double* __restrict__ us_new)
{
    XlFw& F = S.fw;
    const int n = P.n, m = P.m, T = P.T;
    const int lane = threadIdx.x & 63;
    XlFwWave& V = F.wv[threadIdx.x >> 6];
    const double half = PATH ? 1.0 : 0.5;
    double xn = 0.0;
    if (lane < n) {
        xn = x0[lane];
        V.xn[lane] = xn;
        if (xs_new)
            xs_new[lane] = xn;
    }
    xl_wave_sync();
    double J = 0.0;
    for (int i = 0; i < T; ++i) {
        const double* xg = PATH ? P.xg + (int64_t)i * n : P.xg;
        const double* ug = PATH ? P.ug + (int64_t)i * m : P.ug;
        double un = 0.0;
        if (k != nullptr) {
            if (lane < n)
                V.dv[lane] = xn - xs[(int64_t)i * n + lane];
            xl_wave_sync();
            if (lane < m) {
                const double* Kr = K + ((int64_t)i * m + lane) * n;
                double acc = Kr[0] * V.dv[0];
                for (int a = 1; a < n; ++a)
                    acc = fma(Kr[a], V.dv[a], acc);
                un = (us[(int64_t)i * m + lane] + alpha * k[(int64_t)i * m + lane]) + acc;
            }
            xl_wave_sync();
        } else if (lane < m) {
            un = us[(int64_t)i * m + lane];
        }
        if (lane < m) {
            V.un[lane] = un;
            V.ud[lane] = un - ug[lane];
            if (us_new)
                us_new[(int64_t)i * m + lane] = un;
        }
        if (lane < n)
            V.dv[lane] = xn - xg[lane];
        xl_wave_sync();
        // l(x, u) (ilqr_cost.py:532-553; PathQRCost :682-701 without the 1/2)
        double sx = xl_quad(n, V.dv, F.Qr, LDN, V.acc, lane);
        if (!PATH)
            sx = sx * half;
        double su = xl_quad(m, V.ud, P.R, m, V.acc, lane);
        if (!PATH)
            su = su * half;
        J = J + (sx + su);
        // x+ = x + dt ((A x + B u) + c phi(W x))
        double xnext = 0.0;
        if (lane < n) {
            double ax = F.At[0 * LDN + lane] * V.xn[0], z = F.Wt[0 * LDN + lane] * V.xn[0];
            for (int j = 1; j < n; ++j) {
                ax = fma(F.At[j * LDN + lane], V.xn[j], ax);
                z = fma(F.Wt[j * LDN + lane], V.xn[j], z);
            }
            double bu = F.Bt[0 * LDN + lane] * V.un[0];
            for (int j = 1; j < m; ++j)
                bu = fma(F.Bt[j * LDN + lane], V.un[j], bu);
            const double s = 1.0 + z * z;
            const double phi = z / sqrt(s);
            xnext = xn + P.dt * ((ax + bu) + P.sig_c * phi);
        }
        xl_wave_sync();
        if (lane < n) {
            xn = xnext;
            V.xn[lane] = xn;
            if (xs_new)
                xs_new[(int64_t)(i + 1) * n + lane] = xn;
        }
        xl_wave_sync();
    }
    {
        const double* xg = PATH ? P.xg + (int64_t)T * n : P.xg;
        if (lane < n)
            V.dv[lane] = xn - xg[lane];
        xl_wave_sync();
        double sx = xl_quad(n, V.dv, P.Qt, n, V.acc, lane);
        if (!PATH)
            sx = sx * half;
        J = J + sx;
    }
    return J;
}

// ---- stage kernels -----------------------------------------------------------------------------------------

template <bool PATH>
__global__ __launch_bounds__(256) void k_xl_rollout_cost(XlP Pb, int64_t B, const double* x0, const double* us, double* xs,
                                                         double* J)
{
    __shared__ XlMem S;
    const int64_t b = blockIdx.x;
    const XlP P = xl_for_problem(Pb, b);
    xl_forward_load(P, S);
    if ((threadIdx.x >> 6) == 0) {
        const double Jv = xl_forward<PATH>(P, S, nullptr, us + b * P.T * P.m, nullptr, nullptr, x0 + b * P.n, 0.0,
                                           xs + b * (P.T + 1) * P.n, nullptr);
        if (threadIdx.x == 0)
            J[b] = Jv;
    }
}

template <bool PATH>
__global__ __launch_bounds__(256) void k_xl_backward(XlP Pb, XlQP QO, int64_t B, const double* xs, const double* us,
                                                     const double* mu, double* k, double* K, double* dJ, int32_t* info,
                                                     int32_t* qp_result, int32_t* free_mask)
{
    __shared__ XlMem S;
    const int64_t b = blockIdx.x;
    const XlP P = xl_for_problem(Pb, b);
    const int64_t n = P.n, m = P.m, T = P.T;
    XlBwOut o;
    xl_backward<PATH>(P, QO, S, xs + b * (T + 1) * n, us + b * T * m, mu[b], k + b * T * m, K + b * T * m * n, o,
                      qp_result ? qp_result + b * T : nullptr, free_mask ? free_mask + b * T * m : nullptr);
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

struct XlAlphas {
    double a[32];
};

template <bool PATH>
__global__ __launch_bounds__(256) void k_xl_forward_ls(XlP Pb, XlAlphas al, int n_alpha, int store_idx, int64_t B,
                                                       const double* xs, const double* us, const double* k,
                                                       const double* K, double* J_new, double* xs_new, double* us_new)
{
    __shared__ XlMem S;
    const int64_t b = blockIdx.x;
    const XlP P = xl_for_problem(Pb, b);
    const int64_t n = P.n, m = P.m, T = P.T;
    const int wave = threadIdx.x >> 6;
    xl_forward_load(P, S);
    for (int j = wave; j < n_alpha; j += 4) {
        const bool st = j == store_idx;
        const double Jv = xl_forward<PATH>(P, S, xs + b * (T + 1) * n, us + b * T * m, k + b * T * m, K + b * T * m * n,
                                           xs + b * (T + 1) * n, al.a[j], st ? xs_new + b * (T + 1) * n : nullptr,
                                           st ? us_new + b * T * m : nullptr);
        if ((threadIdx.x & 63) == 0)
            J_new[b * n_alpha + j] = Jv;
    }
}

// ---- whole solve --------------------------------------------------------------------------------------------

struct XlSolveBuf { // per resident workgroup: four candidate trajectories and the second copy of the gains
    double *xs, *us, *k, *K;
};

// iLQR.update_u (ilqr_core.py:241-399) for problem b, run by one workgroup.  The line search rolls out four step
// sizes at a time, one per wavefront, and scans their costs in the reference's order (first hit).
template <bool PATH>
__device__ void xl_solve_one(const XlP& P_batch, const XlSolveOpts& O, XlMem& S, const int64_t b, const int64_t B,
                             const double* x0, const double* us_init, double* delta_io, double* k0, double* K0,
                             double* xs0, double* us0, const XlSolveBuf& Wb, double* J_out, double* mu_out,
                             int32_t* iters_out, int32_t* status_out, const ilqr_trace& tr)
{
    const XlP P = xl_for_problem(P_batch, b);
    const int64_t n = P.n, m = P.m, T = P.T;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t sx = (T + 1) * n, su = T * m, sK = T * m * n;
    // trajectory buffers: 0 = the caller's arrays, 1..4 = the workgroup's
    auto XS = [&](int idx) { return idx == 0 ? xs0 + b * sx : Wb.xs + (int64_t)(idx - 1) * sx; };
    auto US = [&](int idx) { return idx == 0 ? us0 + b * su : Wb.us + (int64_t)(idx - 1) * su; };
    auto KS = [&](int idx) { return idx == 0 ? k0 + b * su : Wb.k; };
    auto KK = [&](int idx) { return idx == 0 ? K0 + b * sK : Wb.K; };
    int cur = 0, gcur = 0;
    // us = u_seq_init.copy() ; xs = rollout ; J_opt   (:269-271)
    for (int64_t i = tid; i < su; i += 256)
        US(0)[i] = us_init[b * su + i];
    xl_forward_load(P, S);
    if (wave == 0) {
        const double Jv = xl_forward<PATH>(P, S, nullptr, US(0), nullptr, nullptr, x0 + b * n, 0.0, XS(0), nullptr);
        if (tid == 0)
            S.Jw[0] = Jv;
    }
    __syncthreads();
    double J = S.Jw[0];
    double mu = O.mu_from_array ? mu_out[b] : 1.0; // reset per call (:264); delta_ persists (:62)
    double delta = delta_io[b];
    int status = ILQR_RAN_OUT, iters = 0;
    double dJ1 = 0.0, dJ2 = 0.0, actual = 0.0; // carried over iterations, as in the reference
    if (O.mu_from_array && O.dJ_carry != nullptr) {
        dJ1 = O.dJ_carry[b];
        dJ2 = O.dJ_carry[B + b];
    }
    __syncthreads();

    for (int it = 0; it < O.max_iter; ++it) {
        XlBwOut bw;
        // backward pass with the LinAlgError retry loop (:287-299); when it gives up at mu_max the line search
        // below still runs on the previous gains and Delta_J terms, as in the reference
        for (;;) {
            xl_backward<PATH>(P, O.qp, S, XS(cur), US(cur), mu, KS(gcur ^ 1), KK(gcur ^ 1), bw, nullptr, nullptr);
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
        // line search (:301-313)
        int accepted = 0, j = 0, win = -1;
        double Jnew = 0.0;
        bool ended = false;
        xl_forward_load(P, S);
        for (int pass = 0; pass * 4 < O.n_alpha && !ended; ++pass) {
            const int jw = pass * 4 + wave;
            const int cand = wave < cur ? wave : wave + 1; // the wavefront's candidate buffer: none is `cur`
            if (jw < O.n_alpha) {
                const double Jv = xl_forward<PATH>(P, S, XS(cur), US(cur), KS(gcur), KK(gcur), XS(cur), O.alphas[jw],
                                                   XS(cand), US(cand));
                if ((tid & 63) == 0)
                    S.Jw[wave] = Jv;
            }
            __syncthreads();
            for (int c = 0; c < 4 && pass * 4 + c < O.n_alpha; ++c) {
                j = pass * 4 + c;
                const double alpha = O.alphas[j];
                Jnew = S.Jw[c];
                const double expected = (alpha * dJ1 + alpha * alpha * dJ2) * -1;
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
            if (!ended)
                j = (pass + 1) * 4 < O.n_alpha ? (pass + 1) * 4 : O.n_alpha;
            __syncthreads();
        }
        bool stop = false;
        int converged = 0;
        if (accepted) {
            delta = fmin(1.0, delta) / O.delta0;
            mu = mu * delta;
            if (mu <= O.mu_min)
                mu = 0.0;
            J = Jnew;
            cur = win < cur ? win : win + 1;
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
        if (tid == 0) {
            const int64_t to = (int64_t)it * B + b;
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
    if (cur != 0) {
        for (int64_t i = tid; i < sx; i += 256)
            XS(0)[i] = XS(cur)[i];
        for (int64_t i = tid; i < su; i += 256)
            US(0)[i] = US(cur)[i];
    }
    if (gcur == 1) {
        for (int64_t i = tid; i < su; i += 256)
            KS(0)[i] = KS(1)[i];
        for (int64_t i = tid; i < sK; i += 256)
            KK(0)[i] = KK(1)[i];
    }
    if (tid == 0) {
        delta_io[b] = delta;
        J_out[b] = J;
        mu_out[b] = mu;
        iters_out[b] = iters;
        status_out[b] = status;
        if (O.mu_from_array && O.dJ_carry != nullptr) {
            O.dJ_carry[b] = dJ1;
            O.dJ_carry[B + b] = dJ2;
        }
    }
    __syncthreads();
}

// One problem per workgroup, taken from a queue (a counter in device memory, zeroed by the host before the launch):
// a workgroup that has finished its problem fetches the index of the next unsolved one and leaves when the counter
// has passed B.  Which workgroup solves which problem has no influence on the results.
template <bool PATH>
__global__ __launch_bounds__(256) void k_xl_solve(XlP P, XlSolveOpts O, int64_t B, const double* x0, const double* us_init,
                                                  double* delta_io, double* k0, double* K0, double* xs0, double* us0,
                                                  double* ws, int* queue, double* J_out, double* mu_out,
                                                  int32_t* iters_out, int32_t* status_out, ilqr_trace tr)
{
    __shared__ XlMem S;
    const int64_t n = P.n, m = P.m, T = P.T;
    const int64_t per_group = 4 * ((T + 1) * n + T * m) + T * m + T * m * n;
    XlSolveBuf Wb;
    double* p = ws + (int64_t)blockIdx.x * per_group;
    Wb.xs = p;
    p += 4 * (T + 1) * n;
    Wb.us = p;
    p += 4 * T * m;
    Wb.k = p;
    p += T * m;
    Wb.K = p;
    for (;;) {
        if (threadIdx.x == 0)
            S.next = atomicAdd(queue, 1);
        __syncthreads();
        const int64_t b = S.next;
        __syncthreads();
        if (b >= B)
            break;
        xl_solve_one<PATH>(P, O, S, b, B, x0, us_init, delta_io, k0, K0, xs0, us0, Wb, J_out, mu_out, iters_out,
                           status_out, tr);
    }
}

// ------------------------------------------------------------------ host side

int xl_check(const ilqr_large_model* m)
{
    if (!m)
        return ILQR_ERR_ARG;
    if (m->n < 1 || m->n > XN || m->m < 1 || m->m > XM)
        return ILQR_ERR_UNSUPPORTED;
    if (m->T < 1 || !m->A || !m->B || !m->W || !m->Q || !m->R || !m->Q_terminal || !m->x_goal || !m->u_goal ||
        !m->lower || !m->upper)
        return ILQR_ERR_ARG;
    if (m->cost_kind != ILQR_COST_QR && m->cost_kind != ILQR_COST_PATH_QR)
        return ILQR_ERR_UNSUPPORTED;
    if ((m->goal_stride_x != 0) != (m->goal_stride_u != 0))
        return ILQR_ERR_ARG;
    if (m->goal_stride_x != 0) { // one goal or path per problem
        const bool path = m->cost_kind == ILQR_COST_PATH_QR;
        if (m->goal_stride_x < (path ? ((int64_t)m->T + 1) * m->n : m->n) ||
            m->goal_stride_u < (path ? (int64_t)m->T * m->m : m->m))
            return ILQR_ERR_ARG;
    }
    return ILQR_OK;
}

XlP xl_fill(const ilqr_large_model* m)
{
    XlP P;
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
    P.n = m->n;
    P.m = m->m;
    P.T = m->T;
    P.path = m->cost_kind == ILQR_COST_PATH_QR ? 1 : 0;
    P.gsx = m->goal_stride_x;
    P.gsu = m->goal_stride_u;
    return P;
}

int xl_qp(const ilqr_opts* o, XlQP& q)
{
    if (!o || o->qp_max_iter < 0 || !(o->qp_step_dec > 0.0) || !(o->qp_step_dec < 1.0) || !(o->qp_min_step > 0.0))
        return ILQR_ERR_ARG;
    // the Armijo loop must end: the step falls below min_step after a bounded number of reductions
    volatile double step = 1.0;
    int n = 0;
    do {
        step = step * o->qp_step_dec;
        ++n;
    } while (!(step < o->qp_min_step) && n <= 65536);
    if (n > 65536)
        return ILQR_ERR_UNSUPPORTED;
    q.max_iter = o->qp_max_iter;
    q.min_grad = o->qp_min_grad;
    q.min_rel = o->qp_min_rel_improve;
    q.step_dec = o->qp_step_dec;
    q.min_step = o->qp_min_step;
    q.armijo = o->qp_armijo;
    return ILQR_OK;
}

inline hipStream_t xs_(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int xrc(hipError_t e) { return e == hipSuccess ? ILQR_OK : ILQR_ERR_HIP; }

int64_t xl_groups(int64_t B)
{
    // one workgroup (its LDS holds one problem's matrices) per CU is resident at a time
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1)
        n_cu = 256;
    return B < n_cu ? B : n_cu;
}

} // namespace

extern "C" {

int ilqr_xl_rollout_cost(const ilqr_large_model* model, int64_t B, const double* x0, const double* us, double* xs,
                         double* J, void* stream)
{
    int rc = xl_check(model);
    if (rc != ILQR_OK)
        return rc;
    if (B < 1 || B > 0x7fffffffLL || !x0 || !us || !xs || !J)
        return ILQR_ERR_ARG;
    if (model->cost_kind == ILQR_COST_PATH_QR)
        hipLaunchKernelGGL(k_xl_rollout_cost<true>, dim3((unsigned)B), dim3(256), 0, xs_(stream), xl_fill(model), B, x0,
                           us, xs, J);
    else
        hipLaunchKernelGGL(k_xl_rollout_cost<false>, dim3((unsigned)B), dim3(256), 0, xs_(stream), xl_fill(model), B, x0,
                           us, xs, J);
    return xrc(hipGetLastError());
}

int ilqr_xl_backward(const ilqr_large_model* model, const ilqr_opts* opts, int64_t B, const double* xs,
                     const double* us, const double* mu, double* k, double* K, double* dJ, int32_t* info,
                     int32_t* qp_result, int32_t* free_mask, void* stream)
{
    int rc = xl_check(model);
    if (rc != ILQR_OK)
        return rc;
    XlQP q;
    rc = xl_qp(opts, q);
    if (rc != ILQR_OK)
        return rc;
    if (B < 1 || B > 0x7fffffffLL || !xs || !us || !mu || !k || !K || !dJ || !info)
        return ILQR_ERR_ARG;
    if (model->cost_kind == ILQR_COST_PATH_QR)
        hipLaunchKernelGGL(k_xl_backward<true>, dim3((unsigned)B), dim3(256), 0, xs_(stream), xl_fill(model), q, B, xs,
                           us, mu, k, K, dJ, info, qp_result, free_mask);
    else
        hipLaunchKernelGGL(k_xl_backward<false>, dim3((unsigned)B), dim3(256), 0, xs_(stream), xl_fill(model), q, B, xs,
                           us, mu, k, K, dJ, info, qp_result, free_mask);
    return xrc(hipGetLastError());
}

int ilqr_xl_forward_ls(const ilqr_large_model* model, int64_t B, int32_t n_alpha, const double* alphas,
                       int32_t store_idx, const double* xs, const double* us, const double* k, const double* K,
                       double* J_new, double* xs_new, double* us_new, void* stream)
{
    int rc = xl_check(model);
    if (rc != ILQR_OK)
        return rc;
    if (B < 1 || B > 0x7fffffffLL || n_alpha < 1 || n_alpha > 32 || !alphas || !xs || !us || !k || !K || !J_new)
        return ILQR_ERR_ARG;
    if (store_idx >= n_alpha || (store_idx >= 0 && (!xs_new || !us_new)))
        return ILQR_ERR_ARG;
    XlAlphas al;
    for (int i = 0; i < 32; ++i)
        al.a[i] = i < n_alpha ? alphas[i] : 0.0;
    if (model->cost_kind == ILQR_COST_PATH_QR)
        hipLaunchKernelGGL(k_xl_forward_ls<true>, dim3((unsigned)B), dim3(256), 0, xs_(stream), xl_fill(model), al,
                           (int)n_alpha, (int)store_idx, B, xs, us, k, K, J_new, xs_new, us_new);
    else
        hipLaunchKernelGGL(k_xl_forward_ls<false>, dim3((unsigned)B), dim3(256), 0, xs_(stream), xl_fill(model), al,
                           (int)n_alpha, (int)store_idx, B, xs, us, k, K, J_new, xs_new, us_new);
    return xrc(hipGetLastError());
}

size_t ilqr_xl_workspace_bytes(const ilqr_large_model* model, int64_t B)
{
    if (xl_check(model) != ILQR_OK || B < 1)
        return 0;
    const size_t T = (size_t)model->T, n = (size_t)model->n, m = (size_t)model->m;
    // per resident workgroup: four candidate trajectories and a second copy of the gains; the queue's counter
    return (size_t)xl_groups(B) * sizeof(double) * (4 * ((T + 1) * n + T * m) + T * m + T * m * n) + 256;
}

int ilqr_xl_solve_batched(const ilqr_large_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                          const double* us_init, double* delta, double* k, double* K, double* xs, double* us,
                          double* J, double* mu, int32_t* iters, int32_t* status, const ilqr_trace* trace,
                          void* workspace, size_t workspace_bytes, void* stream)
{
    int rc = xl_check(model);
    if (rc != ILQR_OK)
        return rc;
    if (!opts || !opts->alphas || opts->n_alpha < 0 || opts->n_alpha > 32 || opts->max_iter < 0)
        return ILQR_ERR_ARG;
    if (B < 1 || !x0 || !us_init || !delta || !k || !K || !xs || !us || !J || !mu || !iters || !status || !workspace)
        return ILQR_ERR_ARG;
    if (B > 0x7fffffff - 4096)
        return ILQR_ERR_ARG; // the queue's counter is a 32-bit integer
    const size_t need = ilqr_xl_workspace_bytes(model, B);
    if (workspace_bytes < need)
        return ILQR_ERR_WORKSPACE;
    XlSolveOpts O;
    O.max_iter = opts->max_iter;
    O.n_alpha = opts->n_alpha;
    O.mu_from_array = (opts->mu_init >= 0.0) ? 1 : 0;
    O.tol = opts->tol;
    O.scale = opts->scale;
    O.mu_min = opts->mu_min;
    O.mu_max = opts->mu_max;
    O.delta0 = opts->delta0;
    O.dJ_carry = opts->dJ_carry;
    for (int i = 0; i < 32; ++i)
        O.alphas[i] = i < opts->n_alpha ? opts->alphas[i] : 0.0;
    rc = xl_qp(opts, O.qp);
    if (rc != ILQR_OK)
        return rc;
    const int64_t groups = xl_groups(B);
    int* queue = reinterpret_cast<int*>(static_cast<char*>(workspace) + need - 256);
    hipError_t e = hipMemsetAsync(queue, 0, sizeof(int), xs_(stream));
    if (e != hipSuccess)
        return xrc(e);
    ilqr_trace tr;
    if (trace)
        tr = *trace;
    else
        std::memset(&tr, 0, sizeof(tr));
    double* ws = static_cast<double*>(workspace);
    if (model->cost_kind == ILQR_COST_PATH_QR)
        hipLaunchKernelGGL(k_xl_solve<true>, dim3((unsigned)groups), dim3(256), 0, xs_(stream), xl_fill(model), O, B, x0,
                           us_init, delta, k, K, xs, us, ws, queue, J, mu, iters, status, tr);
    else
        hipLaunchKernelGGL(k_xl_solve<false>, dim3((unsigned)groups), dim3(256), 0, xs_(stream), xl_fill(model), O, B, x0,
                           us_init, delta, k, K, xs, us, ws, queue, J, mu, iters, status, tr);
    return xrc(hipGetLastError());
}

} // extern "C"
