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

// A model of the caller's own on this path: compiled with -DILQR_LARGE_USER_MODEL='"model.h"' this file becomes a
// per-model library (same entry points) whose dynamics are the two functions of the model header -- the contract of
// include/ilqr_user.h: ILQR_USER_N (<= 32 states), ILQR_USER_M (<= 8 controls; u is a pointer when defined),
// ILQR_USER_NP parameters, ilqr_user_f(p, x, u, out), ilqr_user_jac(p, x, u, fx [N*N], fu [N*M]) or ILQR_USER_FD.
// The reference's counterpart is a Dynamics object evaluated once per time step (ilqr/ilqr_dynamics.py:13-91,
// :94-575).  The model's functions are scalar code: a rollout evaluates f on one lane per step size (16 lanes), a
// backward step evaluates the Jacobian on one lane into LDS, from where the matrix products take their tiles; the
// rest of the solver (Q-terms, boxQP, line search) is the MFMA code below.  ilqr_large_model.A then points to the
// parameter vector (device), B and W are not used.
#ifdef ILQR_LARGE_USER_MODEL
#include "ilqr_math.hpp"
#define ILQR_FN __device__ __forceinline__
#define ilqr_sincos(x, s, c) ilqr::det_sincos((x), *(s), *(c))
#define ilqr_atan2(y, x) ilqr::det_atan2((y), (x))
#define ilqr_sqrt(x) sqrt(x)
#include ILQR_LARGE_USER_MODEL
#ifndef ILQR_USER_M
#define ILQR_USER_M 1
#endif
#include "../../include/ilqr_user_fd.h" // ilqr_user_jac by forward differences if the model asks for it
#endif

namespace {

constexpr int LN = 32; // states
constexpr int LM = 8;  // controls
constexpr int LA = 16; // step sizes rolled out together
#ifdef ILQR_LARGE_USER_MODEL
constexpr int UN = ILQR_USER_N, UM = ILQR_USER_M;
static_assert(UN >= 1 && UN <= LN, "ILQR_USER_N must be between 1 and 32 on this path");
static_assert(UM >= 1 && UM <= LM, "ILQR_USER_M must be between 1 and 8 on this path");
#endif

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
    int path; // PathQRCost: xg [T+1][n], ug [T][m] hold the goal of every time step, values carry no 1/2
    int64_t gsx, gsu; // doubles between the goals (or paths) of consecutive problems; 0: shared by the batch
};

// the model as problem b sees it: its own goal or path if the batch has one per problem
__device__ __forceinline__ LargeP for_problem(const LargeP& P, int64_t b)
{
    LargeP Q = P;
    Q.xg = P.xg + b * P.gsx;
    Q.ug = P.ug + b * P.gsu;
    return Q;
}

struct LQPOpts {
    int max_iter;
    int n_exhaust; // number of step reductions after which step < min_step (host-simulated)
    double min_grad, min_rel, step_dec, min_step, armijo;
    // A model with fewer than 8 controls runs zero-padded (see large_check): the padded controls have zero rows in
    // every matrix and the limits [0, 0]; for boxQP they are permanently clamped variables that no chain sees.
    unsigned real_mask; // bit i set: control i exists
    int m_real;
};


struct LSolveOpts {
    int max_iter, n_alpha, mu_from_array;
    double tol, scale, mu_min, mu_max, delta0;
    double alphas[32];
    LQPOpts qp;
    double* dJ_carry; // [2, B] or null: Delta_J1 / Delta_J2 carried from one launch to the next
};

constexpr int LQ_TAB = 512; // Armijo steps tabulated per workgroup

__device__ __forceinline__ double lclamp(double lo, double hi, double v)
{
    // np.maximum(lower, np.minimum(upper, x)) with NaN propagation (ilqr_boxQP.py:37)
    double t = (v != v) ? v : (hi < v ? hi : v);
    return (t != t) ? t : (lo > t ? lo : t);
}

// LDS traffic of one wave needs no workgroup barrier: the wave's LDS instructions execute in
// order; this only keeps the compiler from moving accesses across the hand-over point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------ boxQP of one wavefront
//
// boxQP (ilqr_boxQP.py:9-186) for m = 8, one problem per wavefront.  The parts of a pass that are sums over
// the variables -- H x (the gradient, :82), H_fc x_c (:126-129), the objective (:62, :148), the norm of the
// free gradient (:120), the slope of the Armijo test (:147) -- are fp64 MFMA chains over two k-blocks
// (v_mfma_f64_16x16x4_f64 is the k-ascending chain fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C))))):
// the same terms in the same order as the scalar "first product, then fma" chains, so the same bits.  An
// 8-vector lives in V layout (V8: lane group g holds elements g and g + 4, replicated over the 16 columns),
// the 8 x 8 matrix as two registers of its N-layout tile, and the Armijo loop evaluates 16 consecutive trips
// as the 16 columns of an 8 x 16 candidate matrix.  What is inherently serial -- the Cholesky factor and the
// two triangular solves -- stays one scalar program that every lane runs on wave-uniform values.
// The reference compacts the free variables before factorising and solving (ilqr_boxQP.py:114-132); here
// clamped entries are kept as exact zeros instead: fma(a, 0, s) == s for finite a, so every chain over the
// free (or the clamped) indices has the same terms in the same order and gives the same bits as the compacted
// computation (only the sign of an exact zero can differ).
struct V8 {
    double a, b; // elements g and g + 4
};

__device__ __forceinline__ double rl64(double v, int lane) // v of `lane` (wave-uniform index) on every lane
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// element I of a V8 as a wave-uniform value
template <int I>
__device__ __forceinline__ double v8_uni(const V8& v)
{
    return rl64(I < 4 ? v.a : v.b, 16 * (I & 3));
}
// V8 of a wave-uniform 8-vector
__device__ __forceinline__ V8 v8_of(const double (&u)[LM], int g)
{
    V8 v;
    v.a = g == 0 ? u[0] : (g == 1 ? u[1] : (g == 2 ? u[2] : u[3]));
    v.b = g == 0 ? u[4] : (g == 1 ? u[5] : (g == 2 ? u[6] : u[7]));
    return v;
}
__device__ __forceinline__ V8 v8_load(const double* p, int g) { return V8{p[g], p[g + 4]}; }

// sum over k = 0..7 of X[k][i] Y[k][j] as one chain (two k-blocks): X, Y given by their two registers
__device__ __forceinline__ d4 mf2(double xa, double xb, double ya, double yb)
{
    d4 acc = zero4();
    acc = mfma4(xa, ya, acc);
    acc = mfma4(xb, yb, acc);
    return acc;
}
// a[0] b[0], then fma(a[i], b[i], .) for i = 1..7: every lane gets the sum
__device__ __forceinline__ double v8_dot(const V8& a, const V8& b) { return mf2(a.a, a.b, b.a, b.b)[0]; }

struct H8 {   // Q_uu
    V8 n, t;  // N-layout tiles of H and of H^T (n.a = H[g][c], n.b = H[g+4][c], t.a = H[c][g], t.b = H[c][g+4]; 0 for c >= 8)
    double e; // entry e of the row-major matrix on lane e (for the scalar factorisation: an entry is a lane read)
    __device__ __forceinline__ double operator[](int i) const { return rl64(e, i); }
};
__device__ __forceinline__ H8 h8_load(const double* Hm, LaneId L, int lane) // Hm: row-major 8 x 8 in LDS
{
    H8 H;
    const bool in = L.c < LM;
    const int c = in ? L.c : 0;
    H.n.a = in ? Hm[L.g * LM + c] : 0.0;
    H.n.b = in ? Hm[(L.g + 4) * LM + c] : 0.0;
    H.t.a = in ? Hm[c * LM + L.g] : 0.0;
    H.t.b = in ? Hm[c * LM + L.g + 4] : 0.0;
    H.e = Hm[lane];
    return H;
}
// H x (rows of H times x), V layout
__device__ __forceinline__ V8 h8_mul(const H8& H, const V8& x)
{
    d4 y = mf2(H.t.a, H.t.b, x.a, x.b);
    return V8{y[0], y[1]};
}
// x^T H (x times the columns of H), V layout
__device__ __forceinline__ V8 h8_tmul(const H8& H, const V8& x)
{
    d4 y = mf2(H.n.a, H.n.b, x.a, x.b);
    return V8{y[0], y[1]};
}
// x.g + 0.5 x^T H x as the reference forms it (ilqr_boxQP.py:62)
__device__ __forceinline__ double h8_value(const H8& H, const V8& g, const V8& x)
{
    double xg = v8_dot(x, g);
    double tot = v8_dot(h8_tmul(H, x), x);
    return xg + 0.5 * tot;
}

struct QPR {
    V8 x;                      // in: warm start, out: solution
    const double *g, *lo, *hi; // LDS
    double Lf[LM][LM];         // masked lower factor of H_ff (all ones until the first factorisation, :36)
    double rd[LM];             // 1 / Lf[i][i], formed once per factorisation (the bits every solve would form)
    unsigned fmask;            // free set Lf was computed for (bit i = variable i free)
    bool factored;             // false: boxQP has not factorised yet, H_ff is still the all-ones matrix of :36
    unsigned frm;              // free set of the last classification (wave-uniform)
    int result, n_iter, n_factor, n_bt, nf;
    bool linalg;
};

// Lower Cholesky factor of H restricted to the free set fm (wave-uniform: clamped rows and columns are
// skipped by scalar branches); returns true if a free pivot is not > 0.
template <class HT>
__device__ __forceinline__ bool lr_chol(const HT& H, unsigned fm, double (&Lf)[LM][LM], double (&rd)[LM])
{
    bool fail = false;
#pragma unroll
    for (int j = 0; j < LM; ++j) {
        if (!(fm & (1u << j))) { // clamped column: unit pivot, zero column
            Lf[j][j] = 1.0;
            rd[j] = 1.0;
#pragma unroll
            for (int i = j + 1; i < LM; ++i)
                Lf[i][j] = 0.0;
            continue;
        }
        double s = H[j * LM + j];
#pragma unroll
        for (int k = 0; k < j; ++k)
            s = fma(-Lf[j][k], Lf[j][k], s);
        fail = fail || !(s > 0.0);
        double d = sqrt(s);
        double r = 1.0 / d;
        Lf[j][j] = d;
        rd[j] = r;
#pragma unroll
        for (int i = j + 1; i < LM; ++i) {
            double t = H[i * LM + j];
#pragma unroll
            for (int k = 0; k < j; ++k)
                t = fma(-Lf[i][k], Lf[j][k], t);
            Lf[i][j] = (fm & (1u << i)) ? t * r : 0.0;
        }
    }
    return fail;
}

// cho_solve on the free set, b = 0 on clamped entries on exit
__device__ __forceinline__ void lr_solve(const double (&Lf)[LM][LM], const double (&rd)[LM], unsigned fm,
                                         double (&b)[LM])
{
#pragma unroll
    for (int i = 0; i < LM; ++i) {
        double s = b[i];
#pragma unroll
        for (int j = 0; j < i; ++j)
            s = fma(-Lf[i][j], b[j], s);
        b[i] = (fm & (1u << i)) ? s * rd[i] : 0.0;
    }
#pragma unroll
    for (int i = LM - 1; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int j = i + 1; j < LM; ++j)
            s = fma(-Lf[j][i], b[j], s);
        b[i] = (fm & (1u << i)) ? s * rd[i] : 0.0;
    }
}

// cho_solve with the all-ones "factor" H_ff = np.ones((n, n)) of ilqr_boxQP.py:36 (used until the
// first factorisation, which can only be while nothing is clamped); `real`: the controls that exist (a model
// with fewer than 8 runs zero-padded, and the reference's matrix of ones has the size of the real problem)
__device__ __forceinline__ void lr_solve_ones(double (&b)[LM], unsigned real)
{
#pragma unroll
    for (int i = 0; i < LM; ++i) {
        double s = b[i];
#pragma unroll
        for (int j = 0; j < i; ++j)
            s = fma(-1.0, b[j], s);
        b[i] = (real & (1u << i)) ? s * 1.0 : 0.0;
    }
#pragma unroll
    for (int i = LM - 1; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int j = i + 1; j < LM; ++j)
            s = fma(-1.0, b[j], s);
        b[i] = (real & (1u << i)) ? s * 1.0 : 0.0;
    }
}

// the scalar objective on wave-uniform or per-lane vectors (used by the 64-trip rounds of the Armijo loop)
template <class HT>
__device__ __forceinline__ double lr_value(const HT& H, const double* g, const double (&x)[LM])
{
    double xg = x[0] * g[0];
#pragma unroll
    for (int i = 1; i < LM; ++i)
        xg = fma(x[i], g[i], xg);
    double tot = 0.0;
#pragma unroll
    for (int j = 0; j < LM; ++j) {
        double acc = x[0] * H[j];
#pragma unroll
        for (int i = 1; i < LM; ++i)
            acc = fma(x[i], H[i * LM + j], acc);
        tot = (j == 0) ? acc * x[0] : fma(acc, x[j], tot);
    }
    return xg + 0.5 * tot;
}

#ifdef ILQR_LARGE_QP_PROF // diagnostic build only: shader clocks per part of a boxQP pass, summed over waves
static __device__ unsigned long long g_lqp[8];
#define LQP_T(var) long long var = clock64()
#define LQP_ADD(slot, a, b) lqp_acc[slot] += (unsigned long long)((b) - (a))
#else
#define LQP_T(var)
#define LQP_ADD(slot, a, b)
#endif

// bit i of the result = variable i, from the ballots of a per-lane flag on the two registers of a V8
__device__ __forceinline__ unsigned v8_mask(bool fa, bool fb)
{
    unsigned long long ba = __ballot(fa), bb = __ballot(fb);
    unsigned m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        m |= (unsigned)((ba >> (16 * q)) & 1ull) << q;
        m |= (unsigned)((bb >> (16 * q)) & 1ull) << (q + 4);
    }
    return m;
}

// Armijo loop, proof of exhaustion.  Near a solution the reference's backtracking loop (ilqr_boxQP.py:151-159)
// often runs until the step falls below min_step: the search direction is tiny, every candidate's objective
// equals the old one up to rounding, and (vc - old) / (sdotg + 1e-10) stays below the Armijo constant because of
// the 1e-10.  Literal evaluation of those ~480 trips is the most expensive path of a backward step, so before
// it is started the wave tries to PROVE that every trip from step size `st` on continues:
//   |xc_i - x_i| <= E_i = (1 + 1e-7) st |s_i| + 1.2e-16 |x_i|      (two roundings; clamping only shortens it, as
//                                                                   lo <= x <= hi)
//   exact objective v:  |v(xc) - v(x)| <= sum_i E_i G_i + 1/2 sum_ij E_i |H_ij| E_j,   with
//       G_i >= |g_i + 1/2 sum_j (H_ij + H_ji) x_j|,  G_i = 1.01 (|grad_i| + 1.1e-15 (|g_i| + sum_j |H_ij| |x_j|)
//                                                               + 1/2 sum_j |H_ij - H_ji| |x_j|)
//       (computed gradient + its rounding error + the asymmetry of the computed Q_uu)
//   computed objective (an 8-term chain, 8 chains of 8 terms, a chain over their results, one add):
//       |V(z) - v(z)| <= 3e-15 A,   A = sum_i X_i |g_i| + sum_ij X_i |H_ij| X_j,  X_i = |x_i| + E_i >= |z_i|
//   hence |vc - old| <= NUM = 1.001 (sum E G + 1/2 E|H|E + 6e-15 A) + 1e-200, and the computed slope satisfies
//   |sdotg| <= S = 1.001 sum_i E_i |grad_i|, so the denominator is >= 0.999 (1e-10 - S).
// If S < 1e-10 and NUM < armijo * 0.999 (1e-10 - S), the quotient is below armijo at this step size, and -- all
// bounds grow with st -- at every smaller one: the loop ends at trip n_exhaust with result 2 and that trip's
// candidate, which is then evaluated directly.  Any NaN or infinity makes a comparison false (no proof, literal
// evaluation).  The bounds are generous by the factors shown; sums are MFMA chains of non-negative terms.
__device__ __forceinline__ bool armijo_exhausts(const H8& H, const V8& g, const V8& x, const V8& grad,
                                                const V8& search, double st, double armijo)
{
    const V8 ax{fabs(x.a), fabs(x.b)}, ag{fabs(g.a), fabs(g.b)}, agr{fabs(grad.a), fabs(grad.b)};
    const V8 E{1.0000001 * st * fabs(search.a) + 1.2e-16 * ax.a, 1.0000001 * st * fabs(search.b) + 1.2e-16 * ax.b};
    const V8 X{ax.a + E.a, ax.b + E.b};
    const double hta = fabs(H.t.a), htb = fabs(H.t.b); // |H| with rows on the k axis: sum_j |H_ij| v_j
    const d4 hax = mf2(hta, htb, ax.a, ax.b);
    const d4 asy = mf2(fabs(H.t.a - H.n.a), fabs(H.t.b - H.n.b), ax.a, ax.b);
    const V8 G{1.01 * (agr.a + 1.1e-15 * (ag.a + hax[0]) + 0.5 * asy[0]),
               1.01 * (agr.b + 1.1e-15 * (ag.b + hax[1]) + 0.5 * asy[1])};
    const d4 hE = mf2(hta, htb, E.a, E.b);
    const d4 hX = mf2(hta, htb, X.a, X.b);
    const double d1 = v8_dot(E, G) + 0.5 * v8_dot(E, V8{hE[0], hE[1]});
    const double A = v8_dot(X, ag) + v8_dot(X, V8{hX[0], hX[1]});
    const double NUM = 1.001 * (d1 + 6e-15 * A) + 1e-200;
    const double S = 1.001 * v8_dot(E, agr);
    return (S < 1e-10) && (armijo > 0.0) && (NUM < armijo * (0.999 * (1e-10 - S)));
}

// H: Q_uu, finite and positive definite (the caller has checked); steps: the wave's step table.
// On entry Q.x holds the warm start and Q.Lf / Q.rd / Q.fmask the factor of the full H (the caller
// needed it for the warm start): a factorisation for a free set that Lf already belongs to is not
// repeated (same inputs, same bits); the reference's counter still counts it.
// FULL: all 8 controls exist (the masks for padded controls are then compile-time constants and cost nothing).
template <bool FULL>
__device__ void lr_boxqp(const LQPOpts& O, const H8& H, const double* steps, QPR& Q)
{
    const unsigned real_mask = FULL ? 0xffu : O.real_mask;
#ifdef ILQR_LARGE_QP_PROF
    unsigned long long lqp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const LaneId L = lane_id();
    const int ln = threadIdx.x & 63;
    const V8 gV = v8_load(Q.g, L.g), loV = v8_load(Q.lo, L.g), hiV = v8_load(Q.hi, L.g);
    V8 x{lclamp(loV.a, hiV.a, Q.x.a), lclamp(loV.b, hiV.b, Q.x.b)};
    const unsigned pad = ~real_mask & 0xffu; // controls that only pad the problem to 8
    unsigned clm = pad; // clamped set of the last classification (none of the real controls: ilqr_boxQP.py:27-28)
    Q.frm = 0;
    Q.factored = false;
    double value = h8_value(H, gV, x), oldvalue = 0.0;
    int result = -1, n_iter = 0, n_factor = 0, n_bt = 0;
    bool linalg = false;
    const bool tabulated = O.n_exhaust < LQ_TAB;
    for (int iter = 0; iter < O.max_iter; ++iter) {
        if (result != -1)
            break;
        if (iter > 1 && (oldvalue - value) < O.min_rel * fabs(oldvalue) && (oldvalue - value) > 0.0) {
            result = 4;
            break;
        }
        oldvalue = value;
        LQP_T(q0);
        // grad = g + H x ; clamped = (x == lower & grad > 0) | (x == upper & grad < 0)   (:82-89)
        const V8 hx = h8_mul(H, x);
        const V8 grad{gV.a + hx.a, gV.b + hx.b};
        const bool ca = ((x.a == loV.a) && (grad.a > 0.0)) || ((x.a == hiV.a) && (grad.a < 0.0));
        const bool cb = ((x.b == loV.b) && (grad.b > 0.0)) || ((x.b == hiV.b) && (grad.b < 0.0));
        const unsigned ncl = v8_mask(ca, cb) | pad;
        const bool changed = ncl != clm;
        clm = ncl;
        Q.frm = ~clm & 0xffu;
        const int nclamped = __builtin_popcount(clm);
        if (nclamped == LM) {
            result = 6;
            break;
        }
        LQP_T(q1);
        LQP_ADD(0, q0, q1);
        if (iter == 1 || changed) {
            if (Q.frm != Q.fmask) {
                if (lr_chol(H, Q.frm, Q.Lf, Q.rd)) {
                    linalg = true; // sla.cholesky raises LinAlgError
                    break;
                }
                Q.fmask = Q.frm;
            }
            Q.factored = true;
            ++n_factor;
        }
        LQP_T(q2);
        LQP_ADD(1, q1, q2);
        // sla.norm(grad[free]); dnrm2 of a single element is its magnitude
        const V8 gm{ca ? 0.0 : grad.a, cb ? 0.0 : grad.b};
        double gnorm;
        if (LM - nclamped == 1) {
            const V8 one{ca ? 0.0 : 1.0, cb ? 0.0 : 1.0};
            gnorm = fabs(v8_dot(gm, one));
        } else {
            gnorm = sqrt(v8_dot(gm, gm));
        }
        if (gnorm < O.min_grad) {
            result = 5;
            break;
        }
        ++n_iter;
        LQP_T(q3);
        LQP_ADD(2, q2, q3);
        // grad_clamped = g[free] + H_fc x[clamped]; search[free] = -cho_solve(H_ff, .) - x[free]   (:126-132)
        V8 search;
        {
            const V8 xcl{ca ? x.a : 0.0, cb ? x.b : 0.0};
            const V8 hc = h8_mul(H, xcl);
            const V8 rv{ca ? 0.0 : gV.a + hc.a, cb ? 0.0 : gV.b + hc.b};
            LQP_T(q4);
            LQP_ADD(3, q3, q4);
            double rhs[LM] = {v8_uni<0>(rv), v8_uni<1>(rv), v8_uni<2>(rv), v8_uni<3>(rv),
                              v8_uni<4>(rv), v8_uni<5>(rv), v8_uni<6>(rv), v8_uni<7>(rv)};
            if (Q.factored)
                lr_solve(Q.Lf, Q.rd, Q.frm, rhs);
            else
                lr_solve_ones(rhs, real_mask);
            const V8 sol = v8_of(rhs, L.g);
            search.a = ca ? 0.0 : sol.a * -1 - x.a;
            search.b = cb ? 0.0 : sol.b * -1 - x.b;
            LQP_T(q5);
            LQP_ADD(4, q4, q5);
        }
        LQP_T(q5b);

        // Armijo line search (:140-159).  The candidates of the reference's backtracking loop do not depend
        // on each other.  Trips 0..15 are evaluated as the 16 columns of an 8 x 16 candidate matrix (lane
        // (g, c) forms variables g and g + 4 of trip c); the trip of column c has its objective and slope on
        // the lane with g == c & 3.  The first trip at which the loop would have ended -- the test fails, or
        // the step has dropped below min_step (result 2) -- wins.
        bool found = false;
        if (tabulated) {
            const int t = L.c;
            const bool in_range = t <= O.n_exhaust;
            const double st = steps[in_range ? t : 0];
            const V8 xc{lclamp(loV.a, hiV.a, x.a + st * search.a), lclamp(loV.b, hiV.b, x.b + st * search.b)};
            const double pa = (xc.a - x.a) * grad.a, pb = (xc.b - x.b) * grad.b;
            const d4 sd = mf2(pa, pb, 1.0, 1.0);              // row t: sum of the slope terms, in order
            const d4 xg = mf2(xc.a, xc.b, gV.a, gV.b);        // row t: x_c . g
            const d4 ac = mf2(H.n.a, H.n.b, xc.a, xc.b);      // [j][t]: x_c^T H
            const d4 tt = mf2(ac[0], ac[1], xc.a, xc.b);      // [t][t]: (x_c^T H) x_c
            const int rs = L.c >> 2;
            const double tot_t = rs == 0 ? tt[0] : (rs == 1 ? tt[1] : (rs == 2 ? tt[2] : tt[3]));
            const double sd_t = rs == 0 ? sd[0] : (rs == 1 ? sd[1] : (rs == 2 ? sd[2] : sd[3]));
            const double xg_t = rs == 0 ? xg[0] : (rs == 1 ? xg[1] : (rs == 2 ? xg[2] : xg[3]));
            const double vc = xg_t + 0.5 * tot_t;
            const bool cont = (vc - oldvalue) / (sd_t + 1e-10) < O.armijo;
            const bool stop = (L.g == (L.c & 3)) && in_range && (!cont || t == O.n_exhaust);
            const unsigned long long m = __ballot(stop);
            const unsigned m16 = (unsigned)((m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xffffull);
            if (m16 != 0u) {
                const int cs = __ffs((int)m16) - 1;
                n_bt += cs;
                if (cs == O.n_exhaust)
                    result = 2;
                x.a = __shfl(xc.a, (ln & 48) | cs);
                x.b = __shfl(xc.b, (ln & 48) | cs);
                value = rl64(vc, 16 * (cs & 3) + cs);
                found = true;
            } else if (armijo_exhausts(H, gV, x, grad, search, steps[16], O.armijo)) {
                // every trip from 16 on continues: the loop ends at trip n_exhaust with that trip's candidate
                const double se = steps[O.n_exhaust];
                x = V8{lclamp(loV.a, hiV.a, x.a + se * search.a), lclamp(loV.b, hiV.b, x.b + se * search.b)};
                value = h8_value(H, gV, x);
                n_bt += O.n_exhaust;
                result = 2;
                found = true;
            }
        }
        if (!found) {
            // the trips from 16 on (or all of them, one per round on lane 0, when the step table is too short
            // for the options): 64 at a time, lane = trip, each lane runs the reference's scalar arithmetic
            double xu[LM] = {v8_uni<0>(x), v8_uni<1>(x), v8_uni<2>(x), v8_uni<3>(x),
                             v8_uni<4>(x), v8_uni<5>(x), v8_uni<6>(x), v8_uni<7>(x)};
            const double gu[LM] = {v8_uni<0>(grad), v8_uni<1>(grad), v8_uni<2>(grad), v8_uni<3>(grad),
                                   v8_uni<4>(grad), v8_uni<5>(grad), v8_uni<6>(grad), v8_uni<7>(grad)};
            const double su[LM] = {v8_uni<0>(search), v8_uni<1>(search), v8_uni<2>(search), v8_uni<3>(search),
                                   v8_uni<4>(search), v8_uni<5>(search), v8_uni<6>(search), v8_uni<7>(search)};
            double step = 1.0; // only used without the table
            for (int tb = tabulated ? 16 : 0;; tb += 64) {
                int t = tb + ln;
                bool in_range = tabulated ? t <= O.n_exhaust : ln == 0;
                double st = tabulated ? steps[in_range ? t : 0] : step;
                double xc[LM];
#pragma unroll
                for (int i = 0; i < LM; ++i)
                    xc[i] = lclamp(Q.lo[i], Q.hi[i], xu[i] + st * su[i]);
                double sdotg = (xc[0] - xu[0]) * gu[0];
#pragma unroll
                for (int i = 1; i < LM; ++i)
                    sdotg = sdotg + (xc[i] - xu[i]) * gu[i];
                double vc = lr_value(H, Q.g, xc);
                bool cont = (vc - oldvalue) / (sdotg + 1e-10) < O.armijo;
                if (!tabulated) {
                    int trip = tb / 64;
                    bool small = trip > 0 && st < O.min_step;
                    unsigned long long stopm = __ballot(ln == 0 && (!cont || small));
                    if (stopm != 0ull) {
                        n_bt += trip;
                        if (small)
                            result = 2;
#pragma unroll
                        for (int i = 0; i < LM; ++i)
                            xu[i] = rl64(xc[i], 0);
                        value = rl64(vc, 0);
                        break;
                    }
                    step = step * O.step_dec;
                    continue;
                }
                bool stop = in_range && (!cont || t == O.n_exhaust);
                unsigned long long mask = __ballot(stop);
                if (mask != 0ull) {
                    int w = __ffsll((long long)mask) - 1;
                    int tstar = tb + w;
                    n_bt += tstar;
                    if (tstar == O.n_exhaust)
                        result = 2;
#pragma unroll
                    for (int i = 0; i < LM; ++i)
                        xu[i] = rl64(xc[i], w);
                    value = rl64(vc, w);
                    break;
                }
                if (tb + 64 < O.n_exhaust && armijo_exhausts(H, gV, x, grad, search, steps[tb + 64], O.armijo)) {
                    const double se = steps[O.n_exhaust];
                    const V8 xe{lclamp(loV.a, hiV.a, x.a + se * search.a), lclamp(loV.b, hiV.b, x.b + se * search.b)};
                    value = h8_value(H, gV, xe);
                    n_bt += O.n_exhaust;
                    result = 2;
#pragma unroll
                    for (int i = 0; i < LM; ++i)
                        xu[i] = i < 4 ? rl64(xe.a, 16 * (i & 3)) : rl64(xe.b, 16 * (i & 3));
                    break;
                }
            }
            x = v8_of(xu, L.g);
        }
        LQP_T(q6);
        LQP_ADD(5, q5b, q6);
#ifdef ILQR_LARGE_QP_PROF
        lqp_acc[6] += 1;
#endif
    }
#ifdef ILQR_LARGE_QP_PROF
    lqp_acc[7] += 1;
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < 8; ++q)
            atomicAdd(&g_lqp[q], lqp_acc[q]);
#endif
    Q.x = x;
    Q.result = result;
    Q.n_iter = n_iter;
    Q.n_factor = n_factor;
    Q.n_bt = n_bt;
    Q.nf = __builtin_popcount(Q.frm);
    Q.linalg = linalg;
}

// ------------------------------------------------------------------ per-wave LDS

struct WaveMem { // per problem (= per wave)
    double H[LM * LM], g[LM], lo[LM], hi[LM], x[LM]; // Q_uu, Q_u, bounds, boxQP solution
    double Lf[LM][LM];                               // boxQP factor
    double vx[LN];                                   // parked V_x
    alignas(16) double P1[LN][LN + 1];               // parking space for a 32 x 32 tile set
    double Qux[LM][LN + 1];
    double K[LM][LN + 1];
    double Vn[LN][LN + 1];
    double zb[LA][LN + 1], lxb[LA][LN + 1]; // W x and l_x of 16 consecutive steps (backward sweep)
    double fxvx[LN];                        // f_x^T V_x of the step in hand
    double J[LA];
};

struct BlockMem { // per workgroup: the model, shared by its four problems
    double A[LN][LN + 1], W[LN][LN + 1], QQ[LN][LN + 1]; // QQ = Q + Q^T
    double fu[LN][LM + 1];                               // dt B
    double RR[LM][LM];                                   // R + R^T
    double xg[LN], ug[LM], lower[LM], upper[LM];
    double steps[LQ_TAB]; // 1, dec, dec*dec, ... with the Armijo loop's own rounded multiplications
    WaveMem wv[4];
};

__device__ void block_init(BlockMem& M, const LargeP& P, double step_dec)
{
    for (int i = threadIdx.x; i < LN * LN; i += blockDim.x) {
        int r = i / LN, c = i % LN;
#ifndef ILQR_LARGE_USER_MODEL
        M.A[r][c] = P.A[i];
        M.W[r][c] = P.W[i];
#endif
        M.QQ[r][c] = P.Q[i] + P.Q[c * LN + r];
    }
#ifndef ILQR_LARGE_USER_MODEL
    for (int i = threadIdx.x; i < LN * LM; i += blockDim.x)
        M.fu[i / LM][i % LM] = P.dt * P.B[i];
#endif
    for (int i = threadIdx.x; i < LM * LM; i += blockDim.x)
        M.RR[i / LM][i % LM] = P.R[i] + P.R[(i % LM) * LM + i / LM];
    for (int i = threadIdx.x; i < LN; i += blockDim.x)
        M.xg[i] = P.xg[i];
    for (int i = threadIdx.x; i < LM; i += blockDim.x) {
        M.ug[i] = P.ug[i];
        M.lower[i] = P.lower[i];
        M.upper[i] = P.upper[i];
    }
    if (threadIdx.x == 0) {
        double st = 1.0;
        for (int t = 0; t < LQ_TAB; ++t) {
            M.steps[t] = st;
            st = st * step_dec;
        }
    }
    __syncthreads();
}


// ------------------------------------------------------------------ backward pass

// Diagnostic (ilqr_large_debug_profile): per-wave shader-clock totals of the three phases of a
// backward step (matrix part 1, boxQP + gains, matrix part 2; then T, warm start, boxQP proper,
// boxQP iterations, factorisations), 8 uint64 per problem.
static __device__ unsigned long long* g_large_prof = nullptr;

struct BwOut {
    double dJ1, dJ2;
    int stopped_at, qp_iter, qp_backtrack, n_clamped, rc; // rc: 0 ok, 2 LinAlgError
};

// _backward_pass + _compute_Q_terms (ilqr_core.py:109-228) for one problem.
// xs [T+1][32], us [T][8] -> k [T][8], K [T][8][32] (zeros where the sweep does not reach).
template <bool PATH, bool FULL>
__device__ void large_backward(const LargeP& P, const LQPOpts& QO, const BlockMem& M, WaveMem& S, const double* xs,
                               const double* us, double mu, double* k, double* K, BwOut& out, int32_t* qp_result,
                               int32_t* free_mask)
{
    // (a model with fewer than 8 controls runs zero-padded: FULL = false, masks from the options)
    const unsigned real_mask = FULL ? 0xffu : QO.real_mask;
    const int m_real = FULL ? LM : QO.m_real;
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
    wave_sync();

    // the model matrices come from LDS where they are used; only the small pieces stay in registers
    d4 fu[2][1], RR[1][1], xg[2][1], ug[1][1];
#ifndef ILQR_LARGE_USER_MODEL
    fu[0][0] = load_n(&M.fu[0][0], LM + 1, LN, LM, 0, 0, L); // dt B: the same at every step
    fu[1][0] = load_n(&M.fu[0][0], LM + 1, LN, LM, 1, 0, L);
#endif
    RR[0][0] = load_n(&M.RR[0][0], LM, LM, LM, 0, 0, L);
    xg[0][0] = load_v(P.xg, LN, 0, L);
    xg[1][0] = load_v(P.xg, LN, 1, L);
    ug[0][0] = load_v(P.ug, LM, 0, L);
    if (PATH) { // the terminal goal
        xg[0][0] = load_v(P.xg + (size_t)T * LN, LN, 0, L);
        xg[1][0] = load_v(P.xg + (size_t)T * LN, LN, 1, L);
    }

    // V_x, V_xx = terminal l_x, l_xx (:120-121)
    d4 Vx[2][1], Vxx[2][2];
    {
        d4 xd[2][1];
        xd[0][0] = load_v(xs + (size_t)T * LN, LN, 0, L) - xg[0][0];
        xd[1][0] = load_v(xs + (size_t)T * LN, LN, 1, L) - xg[1][0];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                Vxx[rt][ct] = load_n(P.Qt, LN, LN, LN, rt, ct, L) + load_nt(P.Qt, LN, LN, LN, rt, ct, L);
        mm_tn<2, 2, 1>(Vxx, xd, Vx); // l_x[j] = sum_a xd[a] (Q_T + Q_T^T)[a][j]
    }
    out.dJ1 = 0.0;
    out.dJ2 = 0.0;
    out.stopped_at = -1;
    out.qp_iter = out.qp_backtrack = out.n_clamped = 0;
    out.rc = 0;

    unsigned long long* prof = g_large_prof;
    unsigned long long c_p1 = 0, c_qp = 0, c_p2 = 0, c_warm = 0, c_box = 0, c_qpiter = 0, c_fact = 0;
    for (int i = T - 1; i >= 0; --i) {
        long long t_a = prof ? clock64() : 0;
        { // ---- matrix part 1: f_x, then what the boxQP needs (Q_u, Q_ux, Q_uu)
        // W x (for the Jacobian) and l_x of a step depend on the trajectory only, not on the recursion: they are
        // formed for 16 steps at a time -- the 16 states as the columns of a 32 x 16 matrix, one product each
        // instead of sixteen matrix-vector products that use a sixteenth of every MFMA -- and kept in LDS.
        // Column c of the block that starts at step i is step i - c; each column's chains are those of its own
        // matrix-vector product, term for term.
        const int cb = (T - 1 - i) & (LA - 1);
        if (cb == 0) {
            d4 X16[2][1], XD16[2][1];
            const int st = i - L.c > 0 ? i - L.c : 0; // (columns beyond step 0 repeat it; nobody reads them)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * rt + L.g + 4 * r;
                    const double xvv = xs[(size_t)st * LN + row];
                    X16[rt][0][r] = xvv;
                    XD16[rt][0][r] = xvv - (PATH ? P.xg[(size_t)st * LN + row] : xg[rt][0][r]);
                }
#ifndef ILQR_LARGE_USER_MODEL
            {
                d4 WT[2][2], Z16[2][1];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        WT[rt][ct] = load_nt(&M.W[0][0], LN + 1, LN, LN, rt, ct, L);
                mm_tn<2, 2, 1>(WT, X16, Z16);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        S.zb[L.c][16 * rt + L.g + 4 * r] = Z16[rt][0][r];
            }
#endif
            {
                d4 QQ[2][2], LX16[2][1];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        QQ[rt][ct] = load_n(&M.QQ[0][0], LN + 1, LN, LN, rt, ct, L);
                mm_tn<2, 2, 1>(QQ, XD16, LX16); // l_x[j] = sum_a x_diff[a] (Q + Q^T)[a][j]   (ilqr_cost.py:555-569)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        S.lxb[L.c][16 * rt + L.g + 4 * r] = LX16[rt][0][r];
            }
            wave_sync();
        }
        d4 xv[2][1], uv[1][1];
        xv[0][0] = load_v(xs + (size_t)i * LN, LN, 0, L);
        xv[1][0] = load_v(xs + (size_t)i * LN, LN, 1, L);
        uv[0][0] = load_v(us + (size_t)i * LM, LM, 0, L);

#ifdef ILQR_LARGE_USER_MODEL
        // f_x, f_u of the caller's model at (x_i, u_i): one lane evaluates its Jacobian function into LDS (row-major
        // [n][n] and [n][m], as the model writes them), the matrix products take their tiles from there
        d4 fx[2][2];
        store_v(S.vx, LN, 0, xv[0][0], L);
        store_v(S.vx, LN, 1, xv[1][0], L);
        store_v(S.x, LM, 0, uv[0][0], L);
        wave_sync();
        if (lane == 0) {
#if ILQR_USER_M > 1
            ilqr_user_jac(P.A, S.vx, S.x, &S.Vn[0][0], &S.K[0][0]);
#else
            ilqr_user_jac(P.A, S.vx, S.x[0], &S.Vn[0][0], &S.K[0][0]);
#endif
        }
        wave_sync();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                fx[rt][ct] = load_n(&S.Vn[0][0], UN, UN, UN, rt, ct, L);
            fu[rt][0] = load_n(&S.K[0][0], UM, UN, UM, rt, 0, L);
        }
        wave_sync(); // (the two buffers are rewritten below: f_x is parked in S.Vn, the gains go to S.K)
#else
        // z = W x, g_i = c phi'(z_i), f_x = I + dt (A + diag(g) W)
        d4 fx[2][2];
        {
            d4 z[2][1];
            z[0][0] = load_v(S.zb[cb], LN, 0, L);
            z[1][0] = load_v(S.zb[cb], LN, 1, L);
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
                    d4 An = load_n(&M.A[0][0], LN + 1, LN, LN, rt, ct, L), Wn = load_n(&M.W[0][0], LN + 1, LN, LN, rt, ct, L);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double id = (16 * rt + L.g + 4 * r == 16 * ct + L.c) ? 1.0 : 0.0;
                        fx[rt][ct][r] = id + P.dt * (An[r] + gi[r] * Wn[r]);
                    }
                }
            }
        }
#endif
        // Q_u = l_u + f_u^T V_x                                   (:207, ilqr_cost.py:571-587)
        d4 Qu[1][1];
        {
            d4 ud[1][1], lu[1][1];
            if (PATH) { // the goals of step i (ilqr_cost.py:682-701)
                xg[0][0] = load_v(P.xg + (size_t)i * LN, LN, 0, L);
                xg[1][0] = load_v(P.xg + (size_t)i * LN, LN, 1, L);
                ug[0][0] = load_v(P.ug + (size_t)i * LM, LM, 0, L);
            }
            ud[0][0] = uv[0][0] - ug[0][0];
            mm_tn<1, 1, 1, 2>(RR, ud, lu);
            mm_tn<2, 1, 1>(fu, Vx, Qu);
            Qu[0][0] = lu[0][0] + Qu[0][0];
        }
        // reg = V_xx + mu I ; M2^T = reg^T f_u ; Q_ux = M2 f_x ; Q_uu = l_uu + M2 f_u   (:210-212)
        // Q_ux is formed transposed, as f_x^T [M2^T | V_x]: the same products term for term (a b = b a), and the
        // ninth column of the same 16 MFMAs is f_x^T V_x, the product Q_x needs (:206)
        d4 QuxT[2][1], Quu[1][1];
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
            mm_tn<2, 1, 1>(M2T, fu, Quu);
            Quu[0][0] = RR[0][0] + Quu[0][0];
            d4 Y[2][1];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Y[rt][0][r] = L.c == LM ? Vx[rt][0][r] : M2T[rt][0][r];
            mm_tn<2, 2, 1>(fx, Y, QuxT);
        }
        // ---- hand Q_uu, Q_u, Q_ux and the bounds to the scalar part through LDS, and park f_x, V_xx,
        // V_x there as well: the boxQP below needs the registers, and what it does not need
        // (Q_x, Q_xx) is computed after it
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            store_v(S.vx, LN, rt, Vx[rt][0], L);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                store_n(&S.Vn[0][0], LN + 1, LN, LN, rt, ct, fx[rt][ct], L);
                store_n(&S.P1[0][0], LN + 1, LN, LN, rt, ct, Vxx[rt][ct], L);
            }
        }
        store_n(S.H, LM, LM, LM, 0, 0, Quu[0][0], L);
        store_v(S.g, LM, 0, Qu[0][0], L);
        store_v(S.lo, LM, 0, load_v(M.lower, LM, 0, L) - uv[0][0], L);
        store_v(S.hi, LM, 0, load_v(M.upper, LM, 0, L) - uv[0][0], L);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * rt + L.g + 4 * r;
                if (L.c < LM)
                    S.Qux[L.c][j] = 0.0 + QuxT[rt][0][r]; // (l_ux is identically zero)
                else if (L.c == LM)
                    S.fxvx[j] = QuxT[rt][0][r];
            }
        wave_sync();
        }
        long long t_b = prof ? clock64() : 0;
        // np.all(np.linalg.eigvals(Q_uu) > 0) else print / warn / break   (:137-140)
        const H8 Hq = h8_load(S.H, L, lane);
        const bool finite = __ballot(!(fabs(Hq.e) < __builtin_inf())) == 0ull;
        if (!finite) {
            out.rc = 2;
            break;
        }
        QPR qr;
        qr.g = S.g;
        qr.lo = S.lo;
        qr.hi = S.hi;
        {
            if (lr_chol(Hq, real_mask, qr.Lf, qr.rd)) {
                out.stopped_at = i;
                break;
            }
            // k[i] = cho_solve((R_Q_uu, True), Q_u): the (positive-sign) boxQP warm start   (:145)
            double ws[LM];
#pragma unroll
            for (int a = 0; a < LM; ++a)
                ws[a] = S.g[a];
            lr_solve(qr.Lf, qr.rd, real_mask, ws);
            qr.x = v8_of(ws, L.g);
            qr.fmask = real_mask;
        }
        long long t_b1 = prof ? clock64() : 0;
        lr_boxqp<FULL>(QO, Hq, M.steps, qr);
        long long t_b2 = prof ? clock64() : 0;
        if (qr.linalg) {
            out.rc = 2;
            break;
        }
        out.qp_iter += qr.n_iter;
        out.qp_backtrack += qr.n_bt;
        out.n_clamped += qr.nf < m_real ? 1 : 0;
        if (qp_result && lane == 0)
            qp_result[i] = qr.result;
        if (free_mask && lane == 0) {
#pragma unroll
            for (int a = 0; a < LM; ++a)
                free_mask[(size_t)i * LM + a] = (qr.frm >> a) & 1u;
        }
        // Delta_J1 += k.Q_u ; Delta_J2 += 0.5 k.(Q_uu k)                 (:175-176)
        {
            const V8 gv = v8_load(S.g, L.g);
            const double d1 = v8_dot(qr.x, gv);
            const double d2 = v8_dot(qr.x, h8_mul(Hq, qr.x));
            out.dJ1 = out.dJ1 + d1;
            out.dJ2 = out.dJ2 + d2 * 0.5;
        }
        // K[i, free, :] = -cho_solve((H_ff, True), Q_ux[free, :]); clamped rows stay 0   (:157-160)
        if (lane < LN) {
            double col[LM];
#pragma unroll
            for (int a = 0; a < LM; ++a)
                col[a] = (qr.frm & (1u << a)) ? S.Qux[a][lane] : 0.0;
            if (qr.nf > 0) {
                if (qr.factored)
                    lr_solve(qr.Lf, qr.rd, qr.frm, col);
                else
                    lr_solve_ones(col, real_mask);
            }
#pragma unroll
            for (int a = 0; a < LM; ++a)
                S.K[a][lane] = (qr.frm & (1u << a)) ? col[a] * -1 : 0.0;
        }
        if (L.c == 0) {
            S.x[L.g] = qr.x.a;
            S.x[L.g + 4] = qr.x.b;
        }
        wave_sync();
        long long t_c = prof ? clock64() : 0;
        // ---- back to the matrix part: reload what was parked, then Q_x and Q_xx
        d4 Qx[2][1], Qxx[2][2];
        {
            d4 fx[2][2], QQ[2][2], lx[2][1];
            const int cb = (T - 1 - i) & (LA - 1);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                Vx[rt][0] = load_v(S.vx, LN, rt, L);
                lx[rt][0] = load_v(S.lxb[cb], LN, rt, L); // (formed with the block of 16 steps, above)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    fx[rt][ct] = load_n(&S.Vn[0][0], LN + 1, LN, LN, rt, ct, L);
                    Vxx[rt][ct] = load_n(&S.P1[0][0], LN + 1, LN, LN, rt, ct, L);
                    QQ[rt][ct] = load_n(&M.QQ[0][0], LN + 1, LN, LN, rt, ct, L);
                }
            }
            wave_sync(); // S.Vn is reused below
            // Q_x = l_x + f_x^T V_x          (:206; the product came with Q_ux)
            Qx[0][0] = lx[0][0] + load_v(S.fxvx, LN, 0, L);
            Qx[1][0] = lx[1][0] + load_v(S.fxvx, LN, 1, L);
            // M1^T = V_xx^T f_x ; Q_xx = l_xx + M1 f_x                    (:208)
            d4 M1T[2][2];
            mm_tn<2, 2, 2>(Vxx, fx, M1T);
            mm_tn<2, 2, 2>(M1T, fx, Qxx);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    Qxx[rt][ct] = QQ[rt][ct] + Qxx[rt][ct];
        }
        d4 Qux[1][2], Quu[1][1], Qu[1][1];
        Qux[0][0] = load_n(&S.Qux[0][0], LN + 1, LM, LN, 0, 0, L);
        Qux[0][1] = load_n(&S.Qux[0][0], LN + 1, LM, LN, 0, 1, L);
        Quu[0][0] = load_n(S.H, LM, LM, LM, 0, 0, L);
        Qu[0][0] = load_v(S.g, LM, 0, L);
        d4 Kn[1][2], kv[1][1];
        Kn[0][0] = load_n(&S.K[0][0], LN + 1, LM, LN, 0, 0, L);
        Kn[0][1] = load_n(&S.K[0][0], LN + 1, LM, LN, 0, 1, L);
        kv[0][0] = load_v(S.x, LM, 0, L);
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
            wave_sync();
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    d4 vt = load_nt(&S.Vn[0][0], LN + 1, LN, LN, rt, ct, L);
                    Vxx[rt][ct] = 0.5 * (Vxx[rt][ct] + vt);
                }
        }
        wave_sync(); // the LDS images are rewritten by the next step
        if (prof) {
            long long t_d = clock64();
            c_p1 += t_b - t_a;
            c_qp += t_c - t_b;
            c_warm += t_b1 - t_b;
            c_box += t_b2 - t_b1;
            c_qpiter += qr.n_iter;
            c_fact += qr.n_factor;
            c_p2 += t_d - t_c;
        }
    }
    if (prof && lane == 0) {
        int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        prof[w * 8 + 0] = c_p1;
        prof[w * 8 + 1] = c_qp;
        prof[w * 8 + 2] = c_p2;
        prof[w * 8 + 3] = (unsigned long long)T;
        prof[w * 8 + 4] = c_warm;
        prof[w * 8 + 5] = c_box;
        prof[w * 8 + 6] = c_qpiter;
        prof[w * 8 + 7] = c_fact;
    }
}

// ------------------------------------------------------------------ rollouts

// _forward_pass + _trajectory_cost (ilqr_core.py:85-107, :230-239) for 16 step sizes at once
// (column c of every matrix belongs to alpha[c]); with ROLL the controls are us themselves
// (_dynamics_rollout, :69-83).  J of column c ends up in S.J[c].  The trajectory of column
// `store_col` (if >= 0) is written to xs_new / us_new.
// COST = false: states and controls only (the cost of the stored column is then formed by large_traj_cost).
template <bool ROLL, bool PATH, bool COST = true>
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
#ifndef ILQR_LARGE_USER_MODEL
            WT[rt][ct] = load_nt(P.W, LN, LN, LN, rt, ct, L);
            AT[rt][ct] = load_nt(P.A, LN, LN, LN, rt, ct, L);
#endif
        }
    Rn[0][0] = load_n(P.R, LM, LM, LM, 0, 0, L);
#ifndef ILQR_LARGE_USER_MODEL
    BT[0][0] = load_nt(P.B, LM, LN, LM, 0, 0, L); // B^T [8][32]
    BT[0][1] = load_nt(P.B, LM, LN, LM, 0, 1, L);
#endif
    d4 xg[2], ug = load_v(P.ug, LM, 0, L);
    xg[0] = load_v(P.xg, LN, 0, L);
    xg[1] = load_v(P.xg, LN, 1, L);
    const bool diag_lane[4] = {L.g == L.c, L.g + 4 == L.c, L.g + 8 == L.c, L.g + 12 == L.c};

    d4 Xh[2][1];
    Xh[0][0] = load_v(x0, LN, 0, L);
    Xh[1][0] = load_v(x0, LN, 1, L);
    double J = 0.0; // meaningful on the diagonal lanes (row == column) of tile (0, 0)
    // The operands of a step (old u, x, k, K) are loaded one step ahead: a step's first product waits for them,
    // and nothing else of the step can start before it.
    struct StepOps {
        d4 uv, xo0, xo1, K0, K1, kv;
    };
    auto fetch = [&](int t) {
        StepOps o;
        o.uv = load_v(us + (size_t)t * LM, LM, 0, L);
        if (!ROLL) {
            o.xo0 = load_v(xs + (size_t)t * LN, LN, 0, L);
            o.xo1 = load_v(xs + (size_t)t * LN, LN, 1, L);
            o.K0 = load_nt(K + (size_t)t * LM * LN, LN, LM, LN, 0, 0, L);
            o.K1 = load_nt(K + (size_t)t * LM * LN, LN, LM, LN, 1, 0, L);
            o.kv = load_v(k + (size_t)t * LM, LM, 0, L);
        }
        return o;
    };
    StepOps nxt = fetch(0);
    for (int t = 0; t < T; ++t) {
        const StepOps op = nxt;
        if (t + 1 < T)
            nxt = fetch(t + 1);
        d4 Uh[1][1];
        const d4 uv = op.uv;
        if (ROLL) {
            Uh[0][0] = uv;
        } else {
            // u_new = u + alpha k + K (x_new - x)   (:104, never clamped)
            d4 dX[2][1], KT[2][1], acc[1][1];
            dX[0][0] = Xh[0][0] - op.xo0;
            dX[1][0] = Xh[1][0] - op.xo1;
            KT[0][0] = op.K0;
            KT[1][0] = op.K1;
            mm_tn<2, 1, 1>(KT, dX, acc);
            Uh[0][0] = (uv + alpha_c * op.kv) + acc[0][0];
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
        if (COST) {
            d4 xd[2][1], ud[1][1], ax[2][1], au[1][1], sx[1][1], su[1][1];
            if (PATH) {
                xg[0] = load_v(P.xg + (size_t)t * LN, LN, 0, L);
                xg[1] = load_v(P.xg + (size_t)t * LN, LN, 1, L);
                ug = load_v(P.ug + (size_t)t * LM, LM, 0, L);
            }
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
                    J = J + (PATH ? sx[0][0][r] + su[0][0][r] : sx[0][0][r] * 0.5 + su[0][0][r] * 0.5);
        }
#ifdef ILQR_LARGE_USER_MODEL
        // x+ = f(x, u) of the caller's model: the 16 candidate states go through LDS as 16 plain vectors, lane c
        // evaluates the model for candidate c, and the results return to the matrix layout
        {
            double* xb = &S.P1[0][0];  // [16][LN]
            double* ub = &S.Qux[0][0]; // [16][LM]
            double* ob = &S.Vn[0][0];  // [16][LN]
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xb[L.c * LN + 16 * rt + L.g + 4 * r] = Xh[rt][0][r];
#pragma unroll
            for (int r = 0; r < 2; ++r)
                ub[L.c * LM + L.g + 4 * r] = Uh[0][0][r];
            wave_sync();
            if (lane < LA) {
#if ILQR_USER_M > 1
                ilqr_user_f(P.A, xb + lane * LN, ub + lane * LM, ob + lane * LN);
#else
                ilqr_user_f(P.A, xb + lane * LN, ub[lane * LM], ob + lane * LN);
#endif
            }
            wave_sync();
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * rt + L.g + 4 * r;
                    Xh[rt][0][r] = row < UN ? ob[L.c * LN + row] : 0.0;
                }
            wave_sync(); // the buffers are rewritten by the next step
        }
#else
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
#endif
    }
    if (store_col >= 0 && L.c == store_col) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                xs_new[(size_t)T * LN + 16 * rt + L.g + 4 * r] = Xh[rt][0][r];
    }
    if (COST) {
        d4 xd[2][1], ax[2][1], sx[1][1];
        if (PATH) {
            xg[0] = load_v(P.xg + (size_t)T * LN, LN, 0, L);
            xg[1] = load_v(P.xg + (size_t)T * LN, LN, 1, L);
        }
        xd[0][0] = Xh[0][0] - xg[0];
        xd[1][0] = Xh[1][0] - xg[1];
        mm_tn<2, 2, 1>(Qtn, xd, ax);
        mm_tn<2, 1, 1>(ax, xd, sx);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (diag_lane[r]) {
                J = J + (PATH ? sx[0][0][r] : sx[0][0][r] * 0.5);
                S.J[L.c] = J;
            }
    }
    wave_sync();
}

#ifndef ILQR_LARGE_USER_MODEL
// _forward_pass (ilqr_core.py:85-107) for ONE step size, without costs: what the line search runs first (its first
// step size is accepted in > 90 % of the iterations).  A rollout of a single candidate is a chain of matrix-VECTOR
// products, for which the 16-column products of large_forward pay sixteen times: here lane r < 32 holds row r of A
// in registers, lane 32 + r row r of W, and every lane forms its row's 32-term chain acc = fma(M[r][k], x[k], acc),
// k ascending from acc = 0 -- term for term the chain the matrix instruction forms for column 0 (so the same bits);
// beside it runs the chain of row (lane & 7) of K_t (x_new - x).  The state and its difference to the old
// trajectory go through LDS (every lane reads all 32 of each, one address per read); K_t, and per lane its elements
// of the old controls, k_t and the old next state, are fetched RL_D steps ahead into registers (K_t then passes
// through LDS, one step ahead, so that a lane can read its row).  u = (u_old + alpha k) + K dx is the same on all
// lanes with the same (lane & 7); B u takes it from lanes 0..7; c phi(W x) moves from lane 32 + r to lane r.
constexpr int RL_D = 6; // (even: the LDS buffers alternate with the step's parity)

struct alignas(16) RollMem { // lives in WaveMem::P1, which is free outside the backward sweep
    double xh[2][LN], dxh[2][LN];
    double Kb[2][LM * (LN + 2)]; // rows 34 doubles apart: 16-byte aligned, and 8 rows on 8 x 4 different banks
};

typedef double d2 __attribute__((ext_vector_type(2)));
// The trajectory and gain arrays reach the solver through run-time selected pointers, which the compiler can only
// treat as generic: generic loads count as LDS operations as well, so every wait for an LDS read would also wait for
// the fetches that are meant to stay in flight for six steps.  Here they are said to be global memory.
typedef const double __attribute__((address_space(1))) * GlobalIn;
typedef const d2 __attribute__((address_space(1))) * GlobalIn2;
typedef double __attribute__((address_space(1))) * GlobalOut;

// (not inlined: with registers of its own, free of the solver's, nothing of it is spilled -- a reload from scratch
// in the step would make the wait for it a wait for every fetch in flight)
__device__ __noinline__ void large_forward_first(const LargeP& P, WaveMem& S, const double* xs, const double* us, const double* k,
                                    const double* K, double alpha, double* xs_new, double* us_new)
{
    static_assert(sizeof(RollMem) <= sizeof(S.P1) && RL_D % 2 == 0, "");
    // (S arrives as a generic reference; said to be LDS, the accesses below are LDS instructions)
    typedef RollMem __attribute__((address_space(3))) LdsRollMem;
    LdsRollMem& R = *(LdsRollMem*)(&S.P1[0][0]);
    const int lane = threadIdx.x & 63;
    const int row = lane & 31, a = lane & 7;
    const int T = P.T;
    const GlobalIn xs_g = (GlobalIn)xs, us_g = (GlobalIn)us, k_g = (GlobalIn)k;
    const GlobalIn2 K_g = (GlobalIn2)K;
    const GlobalOut xs_o = (GlobalOut)xs_new, us_o = (GlobalOut)us_new;
    double Mrow[LN], Brow[LM];
    {
        const GlobalIn2 mp = (GlobalIn2)((lane >= 32 ? P.W : P.A) + row * LN);
#pragma unroll
        for (int q = 0; q < LN / 2; ++q) {
            const d2 v = mp[q];
            Mrow[2 * q] = v.x;
            Mrow[2 * q + 1] = v.y;
        }
#pragma unroll
        for (int q = 0; q < LM; ++q)
            Brow[q] = ((GlobalIn)P.B)[row * LM + q];
    }
    struct KStage {
        d2 k0, k1; // elements 4 lane .. 4 lane + 3 of K_t (row-major 8 x 32)
    };
    struct VStage {
        double xo, uo, kk; // x_old[t + 1][row], u_old[t][a], k[t][a]
    };
    KStage Ks[RL_D];
    VStage Vs[RL_D];
    auto fetch_K = [&](KStage& st, int t) {
        const int tc = t < T ? t : T - 1;
        const GlobalIn2 kp = K_g + (size_t)tc * (LM * LN / 2) + 2 * lane;
        st.k0 = kp[0];
        st.k1 = kp[1];
    };
    auto fetch_V = [&](VStage& st, int t) {
        const int tc = t < T ? t : T - 1;
        st.xo = xs_g[(size_t)(tc + 1) * LN + row];
        st.uo = us_g[(size_t)tc * LM + a];
        st.kk = k_g[(size_t)tc * LM + a];
    };
    auto put_K = [&](const KStage& st, int buf) { // element e = 4 lane of the 8 x 32 matrix -> row e / 32, column e % 32
        typedef d2 __attribute__((address_space(3))) LdsD2;
        LdsD2* dst = (LdsD2*)(&R.Kb[buf][(lane >> 3) * (LN + 2) + 4 * (lane & 7)]);
        dst[0] = st.k0;
        dst[1] = st.k1;
    };
    double xme = xs_g[row]; // x_new[0] = x_old[0]
    const double dt = P.dt, sig_c = P.sig_c;
    // Everything loaded so far is waited for HERE (vmcnt(0)), before the fetches that stay in flight are issued:
    // a value still pending at the head of the step loop would be waited for in every step, and a wait for the
    // oldest load is a wait for all of them.
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int j = 0; j < RL_D; ++j) {
        fetch_K(Ks[j], j);
        fetch_V(Vs[j], j);
    }
    wave_sync(); // (the backward sweep's last use of this LDS is over)
    if (lane < LN) {
        R.xh[0][row] = xme;
        R.dxh[0][row] = xme - xs_g[row];
        xs_o[row] = xme;
    }
    put_K(Ks[0], 0);
    fetch_K(Ks[0], RL_D);
    wave_sync();
    auto step = [&](int t, int par, KStage& kn, VStage& vs) {
        // K of the next step into the other buffer, and the fetch of the step RL_D after it
        put_K(kn, par ^ 1);
        fetch_K(kn, t + 1 + RL_D);
        typedef const d2 __attribute__((address_space(3))) LdsD2In;
        LdsD2In* xv = (LdsD2In*)(R.xh[par]);
        LdsD2In* dv = (LdsD2In*)(R.dxh[par]);
        LdsD2In* kr = (LdsD2In*)(&R.Kb[par][a * (LN + 2)]);
        double accM = 0.0, accK = 0.0; // (A x)[row] or (W x)[row]; (K dx)[a]
#pragma unroll
        for (int q = 0; q < LN / 2; ++q) {
            const d2 x2 = xv[q], dx2 = dv[q], k2 = kr[q];
            accM = fma(Mrow[2 * q], x2.x, accM);
            accK = fma(k2.x, dx2.x, accK);
            accM = fma(Mrow[2 * q + 1], x2.y, accM);
            accK = fma(k2.y, dx2.y, accK);
        }
        // u_new = u + alpha k + K (x_new - x)   (:104, never clamped)
        const double u = (vs.uo + alpha * vs.kk) + accK;
        double bu = 0.0;
#pragma unroll
        for (int q = 0; q < LM; ++q)
            bu = fma(Brow[q], rl64(u, q), bu);
        // x+ = x + dt (A x + B u + c phi(W x)), phi(z) = z / sqrt(1 + z z): formed where W x is (lanes 32..63)
        const double sq = 1.0 + accM * accM;
        const double cphi = sig_c * (accM / sqrt(sq));
        const double cphi_r = __shfl_xor(cphi, 32);
        const double xn = xme + dt * ((accM + bu) + cphi_r);
        if (lane < LN) {
            R.xh[par ^ 1][row] = xn;
            R.dxh[par ^ 1][row] = xn - vs.xo;
            xs_o[(size_t)(t + 1) * LN + row] = xn;
        }
        if (lane < LM)
            us_o[(size_t)t * LM + a] = u;
        xme = xn;
        fetch_V(vs, t + RL_D);
        wave_sync();
    };
    for (int t0 = 0; t0 < T; t0 += RL_D) {
#pragma unroll
        for (int j = 0; j < RL_D; ++j)
            if (t0 + j < T)
                step(t0 + j, j & 1, Ks[(j + 1) % RL_D], Vs[j]);
    }
    wave_sync();
}
#endif

// _trajectory_cost (ilqr_core.py:230-239) of a stored trajectory: the stage costs of 16 time steps at a time, the
// 16 steps as the columns of the same products a rollout forms per step for its 16 step sizes (so each l_t has the
// bits the rollout would give it), then J = J + l_t in the order of t, and the terminal cost.  The same number on
// every lane.  The line search uses it for the first step size, which is accepted in > 90 % of the iterations: the
// rollout then carries no cost products (28 of its 76 MFMAs per step), and the cost costs 28 MFMAs per 16 steps.
template <bool PATH>
__device__ double large_traj_cost(const LargeP& P, WaveMem& S, const double* xs, const double* us)
{
    const LaneId L = lane_id();
    const int T = P.T;
    d4 Qn[2][2], Rn[1][1];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            Qn[rt][ct] = load_n(P.Q, LN, LN, LN, rt, ct, L);
    Rn[0][0] = load_n(P.R, LM, LM, LM, 0, 0, L);
    const d4 ug = load_v(P.ug, LM, 0, L);
    d4 xg[2];
    xg[0] = load_v(P.xg, LN, 0, L);
    xg[1] = load_v(P.xg, LN, 1, L);
    const bool diag_lane[4] = {L.g == L.c, L.g + 4 == L.c, L.g + 8 == L.c, L.g + 12 == L.c};
    double J = 0.0;
    // The stage costs of the whole horizon are kept in LDS (in the space of the backward sweep's 16-step blocks, free
    // here) and summed at the end, so that the rounds of 16 steps follow each other without a barrier; a horizon too
    // long for that space is summed round by round.
    double* const stage = &S.zb[0][0];
    static_assert(sizeof(S.zb) + sizeof(S.lxb) == 2 * LA * (LN + 1) * sizeof(double), "");
    const bool all_in_lds = T + LA <= 2 * LA * (LN + 1);
    for (int t0 = 0; t0 < T; t0 += LA) {
        const int st = t0 + L.c < T ? t0 + L.c : T - 1; // (columns beyond the last step repeat it; they are not added)
        d4 xd[2][1], ud[1][1], ax[2][1], au[1][1], sx[1][1], su[1][1];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + L.g + 4 * r;
                xd[rt][0][r] = xs[(size_t)st * LN + row] - (PATH ? P.xg[(size_t)st * LN + row] : xg[rt][r]);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = L.g + 4 * r;
            ud[0][0][r] = row < LM ? us[(size_t)st * LM + row] - (PATH ? P.ug[(size_t)st * LM + row] : ug[r]) : 0.0;
        }
        mm_tn<2, 2, 1>(Qn, xd, ax);
        mm_tn<2, 1, 1>(ax, xd, sx);
        mm_tn<1, 1, 1, 2>(Rn, ud, au);
        mm_tn<1, 1, 1, 2>(au, ud, su);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (diag_lane[r])
                (all_in_lds ? stage[t0 + L.c] : S.J[L.c]) =
                    PATH ? sx[0][0][r] + su[0][0][r] : sx[0][0][r] * 0.5 + su[0][0][r] * 0.5;
        if (!all_in_lds) {
            wave_sync();
            const int nc = T - t0 < LA ? T - t0 : LA;
            for (int c = 0; c < nc; ++c)
                J = J + S.J[c];
            wave_sync();
        }
    }
    if (all_in_lds) { // the sum over t, in the order of t, once all stage costs are there
        wave_sync();
#pragma unroll 8
        for (int t = 0; t < T; ++t)
            J = J + stage[t];
        wave_sync();
    }
    {
        d4 Qtn[2][2], xd[2][1], ax[2][1], sx[1][1];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                Qtn[rt][ct] = load_n(P.Qt, LN, LN, LN, rt, ct, L);
        if (PATH) {
            xg[0] = load_v(P.xg + (size_t)T * LN, LN, 0, L);
            xg[1] = load_v(P.xg + (size_t)T * LN, LN, 1, L);
        }
        xd[0][0] = load_v(xs + (size_t)T * LN, LN, 0, L) - xg[0];
        xd[1][0] = load_v(xs + (size_t)T * LN, LN, 1, L) - xg[1];
        mm_tn<2, 2, 1>(Qtn, xd, ax);
        mm_tn<2, 1, 1>(ax, xd, sx);
        if (diag_lane[0] && L.c == 0)
            S.J[0] = sx[0][0][0];
        wave_sync();
        const double sT = S.J[0];
        J = J + (PATH ? sT : sT * 0.5);
        wave_sync();
    }
    return J;
}

// ------------------------------------------------------------------ kernels

template <bool PATH>
__global__ __launch_bounds__(256) void k_large_rollout_cost(LargeP P, int64_t B, const double* x0, const double* us,
                                                            double* xs, double* J)
{
    __shared__ BlockMem M;
    block_init(M, P, 0.5);
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B)
        return;
    WaveMem& S = M.wv[threadIdx.x >> 6];
    large_forward<true, PATH>(for_problem(P, b), S, nullptr, us + b * P.T * LM, nullptr, nullptr, x0 + b * LN, 0.0, 0,
                        xs + b * (P.T + 1) * LN, nullptr);
    if ((threadIdx.x & 63) == 0)
        J[b] = S.J[0];
}

template <bool PATH, bool FULL>
__global__ __launch_bounds__(256) void k_large_backward(LargeP P, LQPOpts QO, int64_t B, const double* xs,
                                                        const double* us, const double* mu, double* k, double* K,
                                                        double* dJ, int32_t* info, int32_t* qp_result,
                                                        int32_t* free_mask)
{
    __shared__ BlockMem M;
    block_init(M, P, QO.step_dec);
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B)
        return;
    WaveMem& S = M.wv[threadIdx.x >> 6];
    BwOut o;
    large_backward<PATH, FULL>(for_problem(P, b), QO, M, S, xs + b * (P.T + 1) * LN, us + b * P.T * LM, mu[b], k + b * P.T * LM,
                   K + b * P.T * LM * LN, o, qp_result ? qp_result + b * P.T : nullptr,
                   free_mask ? free_mask + b * P.T * LM : nullptr);
    if ((threadIdx.x & 63) == 0) {
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
template <bool PATH>
__global__ __launch_bounds__(256) void k_large_forward_ls(LargeP P, LAlphas al, int n_alpha, int store_idx, int64_t B,
                                                          const double* xs, const double* us, const double* k,
                                                          const double* K, double* J_new, double* xs_new,
                                                          double* us_new)
{
    __shared__ BlockMem M;
    block_init(M, P, 0.5);
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B)
        return;
    WaveMem& S = M.wv[threadIdx.x >> 6];
    const LaneId L = lane_id();
    const int lane = threadIdx.x & 63;
    for (int pass = 0; pass * LA < n_alpha; ++pass) {
        int j = pass * LA + L.c;
        double a = j < n_alpha ? al.a[j < 32 ? j : 0] : 0.0;
        int sc = (store_idx >= pass * LA && store_idx < (pass + 1) * LA) ? store_idx - pass * LA : -1;
        large_forward<false, PATH>(for_problem(P, b), S, xs + b * (P.T + 1) * LN, us + b * P.T * LM, k + b * P.T * LM,
                             K + b * P.T * LM * LN, xs + b * (P.T + 1) * LN, a, sc,
                             xs_new ? xs_new + b * (P.T + 1) * LN : nullptr, us_new ? us_new + b * P.T * LM : nullptr);
        if (lane < LA && pass * LA + lane < n_alpha)
            J_new[b * n_alpha + pass * LA + lane] = S.J[lane];
        wave_sync();
    }
}

struct LSolveBuf {
    double *xs1, *us1, *k1, *K1; // second copies (workspace); the first copies are the caller's arrays
};

// iLQR.update_u (ilqr_core.py:241-399) for problem b, run by one wavefront.
template <bool PATH, bool FULL>
__device__ __forceinline__ void large_solve_one(const LargeP& P_batch, const LSolveOpts& O, BlockMem& M, WaveMem& S,
                                                const int64_t b, const int64_t B, const double* x0,
                                                const double* us_init, double* delta_io, double* k0, double* K0,
                                                double* xs0, double* us0, const LSolveBuf& W, double* J_out,
                                                double* mu_out, int32_t* iters_out, int32_t* status_out,
                                                const ilqr_trace& tr)
{
    const LargeP P = for_problem(P_batch, b);
    const int T = P.T;
    const LaneId L = lane_id();
    const int lane = threadIdx.x & 63;
    // the two copies of each array: the caller's and the workspace's (selected, not indexed: an array of pointers
    // indexed at run time lives in scratch memory, and what is loaded back from there is no longer known to be global)
    double* const xs_a = xs0 + b * (T + 1) * LN, * const xs_b = W.xs1 + b * (T + 1) * LN;
    double* const us_a = us0 + b * T * LM, * const us_b = W.us1 + b * T * LM;
    double* const k_a = k0 + b * T * LM, * const k_b = W.k1 + b * T * LM;
    double* const K_a = K0 + b * T * LM * LN, * const K_b = W.K1 + b * T * LM * LN;
    auto xsb = [&](int i) { return i ? xs_b : xs_a; };
    auto usb = [&](int i) { return i ? us_b : us_a; };
    auto kb = [&](int i) { return i ? k_b : k_a; };
    auto Kb = [&](int i) { return i ? K_b : K_a; };
    int cur = 0, gcur = 0;
    // us = u_seq_init.copy() ; xs = rollout ; J_opt                     (:269-271)
    for (int i = lane; i < T * LM; i += 64)
        usb(0)[i] = us_init[b * T * LM + i];
    wave_sync();
    large_forward<true, PATH>(P, S, nullptr, usb(0), nullptr, nullptr, x0 + b * LN, 0.0, 0, xsb(0), nullptr);
    double J = S.J[0];
    double mu = O.mu_from_array ? mu_out[b] : 1.0; // reset per call (:264); delta_ persists (:62)
    double delta = delta_io[b];
    int status = ILQR_RAN_OUT, iters = 0;
    double dJ1 = 0.0, dJ2 = 0.0, actual = 0.0; // carried over iterations, as in the reference
    if (O.mu_from_array && O.dJ_carry != nullptr) { // a continued solve: the sums of its last backward pass
        dJ1 = O.dJ_carry[b];
        dJ2 = O.dJ_carry[B + b];
    }
    wave_sync();

    for (int it = 0; it < O.max_iter; ++it) {
        BwOut bw;
        // backward pass with the LinAlgError retry loop (:287-299); when it gives up at mu_max
        // the line search below still runs on the previous gains and Delta_J terms, as in the reference
        for (;;) {
            large_backward<PATH, FULL>(P, O.qp, M, S, xsb(cur), usb(cur), mu, kb(gcur ^ 1), Kb(gcur ^ 1), bw, nullptr, nullptr);
            wave_sync();
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
        if (O.n_alpha > 0) {
            // the first step size, accepted in > 90 % of the iterations: its rollout without the cost products, its
            // cost from the stored trajectory (large_traj_cost: the same bits); only if it is turned down do the
            // candidates get rolled out with their costs, below (the scan then starts again at the first step size)
#ifdef ILQR_LARGE_USER_MODEL
            const double a = L.c < O.n_alpha ? O.alphas[L.c] : 0.0;
            large_forward<false, PATH, false>(P, S, xsb(cur), usb(cur), kb(gcur), Kb(gcur), xsb(cur), a, 0, xsb(nxt),
                                              usb(nxt));
#else
            large_forward_first(P, S, xsb(cur), usb(cur), kb(gcur), Kb(gcur), O.alphas[0], xsb(nxt), usb(nxt));
#endif
            wave_sync();
            Jnew = large_traj_cost<PATH>(P, S, xsb(nxt), usb(nxt));
            const double alpha = O.alphas[0];
            const double expected = (alpha * dJ1 + alpha * alpha * dJ2) * -1;
            if (expected <= 0.0) {
                ended = true; // reject the whole iteration
            } else if ((J - Jnew) / expected > O.scale) {
                actual = J - Jnew;
                accepted = 1;
                ended = true;
            }
        }
        {
            for (int pass = 0; pass * LA < O.n_alpha && !ended; ++pass) {
                int jj = pass * LA + L.c;
                double a = jj < O.n_alpha ? O.alphas[jj < 32 ? jj : 0] : 0.0;
                large_forward<false, PATH>(P, S, xsb(cur), usb(cur), kb(gcur), Kb(gcur), xsb(cur), a, pass == 0 ? 0 : -1,
                                     xsb(nxt), usb(nxt));
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
                wave_sync();
                if (!ended)
                    j = (pass + 1) * LA < O.n_alpha ? (pass + 1) * LA : O.n_alpha;
                if (accepted && !(pass == 0 && win == 0)) {
                    // the accepted candidate's states and controls (same arithmetic, same bits)
                    double aw = O.alphas[pass * LA + L.c < O.n_alpha ? pass * LA + L.c : 0];
                    large_forward<false, PATH>(P, S, xsb(cur), usb(cur), kb(gcur), Kb(gcur), xsb(cur), aw, win, xsb(nxt),
                                         usb(nxt));
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
    wave_sync();
    // results back into the caller's arrays
    if (cur == 1) {
        for (int i = lane; i < (T + 1) * LN; i += 64)
            xsb(0)[i] = xsb(1)[i];
        for (int i = lane; i < T * LM; i += 64)
            usb(0)[i] = usb(1)[i];
    }
    if (gcur == 1) {
        for (int i = lane; i < T * LM; i += 64)
            kb(0)[i] = kb(1)[i];
        for (int i = lane; i < T * LM * LN; i += 64)
            Kb(0)[i] = Kb(1)[i];
    }
    if (lane == 0) {
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
    wave_sync(); // the wave's LDS region is reused by its next problem
}

// One problem per wavefront, taken from a queue: a wavefront that has finished its problem fetches the index
// of the next unsolved one from a counter in device memory (zeroed by the host before the launch) and leaves
// when the counter has passed B.  Iteration counts differ widely between problems (mean 11, maximum 116 on
// the synthetic workload), so a workgroup of four fixed problems would hold its CU until the slowest of the
// four has ended.  Which wavefront solves which problem has no influence on the results.
template <bool PATH, bool FULL>
__global__ __launch_bounds__(256) void k_large_solve(LargeP P, LSolveOpts O, int64_t B, const double* x0,
                                                     const double* us_init, double* delta_io, double* k0, double* K0,
                                                     double* xs0, double* us0, LSolveBuf W, int* queue, double* J_out,
                                                     double* mu_out, int32_t* iters_out, int32_t* status_out,
                                                     ilqr_trace tr)
{
    __shared__ BlockMem M;
    block_init(M, P, O.qp.step_dec);
    WaveMem& S = M.wv[threadIdx.x >> 6];
    for (;;) {
        int next = 0;
        if ((threadIdx.x & 63) == 0)
            next = atomicAdd(queue, 1);
        const int64_t b = __builtin_amdgcn_readfirstlane(next);
        if (b >= B)
            break;
        large_solve_one<PATH, FULL>(P, O, M, S, b, B, x0, us_init, delta_io, k0, K0, xs0, us0, W, J_out, mu_out, iters_out,
                              status_out, tr);
    }
}

// ------------------------------------------------------- derivative tensor stream

// The tensors _compute_Q_terms asks the model and the cost for (ilqr_core.py:189-198) at every (problem, step),
// written as dense streams in the reference's layout: f_x [B,T,n,n], f_u [B,T,n,m], l_x [B,T,n], l_u [B,T,m],
// l_xx [B,T,n,n], l_uu [B,T,m,m], l_ux [B,T,m,n], terminal l_x [B,n], l_xx [B,n,n]: 21.3 KB per (problem,
// step), a pure HBM write stream.  One workgroup handles DS consecutive steps of one problem: the 32-term
// chains (W x, and x_diff^T (Q + Q^T)) run on 32 lanes per step out of LDS copies of W and Q + Q^T, then all
// 256 threads write the element-wise parts, 256 consecutive doubles per store.
//   f_x = I + dt (A + c diag(phi'(W x)) W),  phi'(z) = 1 / ((1 + z^2) sqrt(1 + z^2));   f_u = dt B.
constexpr int DS = 8; // steps per workgroup

template <bool PATH>
__global__ __launch_bounds__(256) void k_large_deriv(LargeP P, int64_t B, const double* __restrict__ xs,
                                                     const double* __restrict__ us, ilqr_deriv_bm_t o)
{
    __shared__ double s_W[LN][LN + 1], s_QQ[LN][LN + 1], s_RR[LM][LM];
    __shared__ double s_x[DS][LN], s_g[DS][LN];
    const int T = P.T;
    const int chunks = (T + 1 + DS - 1) / DS; // step T (terminal) included
    const int64_t b = blockIdx.x / chunks;
    const int t0 = (int)(blockIdx.x % chunks) * DS;
    const int tid = threadIdx.x;
    for (int e = tid; e < LN * LN; e += 256) {
        int i = e / LN, j = e % LN;
        s_W[i][j] = P.W[e];
        s_QQ[i][j] = P.Q[i * LN + j] + P.Q[j * LN + i]; // (Q + Q^T)[i][j], ilqr_cost.py:526
    }
    if (tid < LM * LM)
        s_RR[tid / LM][tid % LM] = P.R[tid] + P.R[(tid % LM) * LM + tid / LM];
    {
        int s = tid >> 5, i = tid & 31, t = t0 + s;
        s_x[s][i] = t <= T ? xs[(b * (T + 1) + t) * LN + i] : 0.0;
    }
    __syncthreads();
    {   // 32-term chains, one (step, row) per thread
        const int s = tid >> 5, i = tid & 31, t = t0 + s;
        if (t < T) {
            double z = s_W[i][0] * s_x[s][0];
#pragma unroll 8
            for (int j = 1; j < LN; ++j)
                z = fma(s_W[i][j], s_x[s][j], z);
            double sq = 1.0 + z * z;
            s_g[s][i] = P.sig_c * (1.0 / (sq * sqrt(sq)));
            // l_x[i] = sum_a x_diff[a] (Q + Q^T)[a][i]   (ilqr_cost.py:555-569)
            const double* xg = P.xg + b * P.gsx + (PATH ? (int64_t)t * LN : 0);
            double acc = (s_x[s][0] - xg[0]) * s_QQ[0][i];
#pragma unroll 8
            for (int a = 1; a < LN; ++a)
                acc = fma(s_x[s][a] - xg[a], s_QQ[a][i], acc);
            o.lx[(b * T + t) * LN + i] = acc;
            if (i < LM) { // l_u[i] = sum_a u_diff[a] (R + R^T)[a][i]   (:571-587)
                const double* ug = P.ug + b * P.gsu + (PATH ? (int64_t)t * LM : 0);
                const double* u = us + (b * T + t) * LM;
                double au = (u[0] - ug[0]) * s_RR[0][i];
#pragma unroll
                for (int a = 1; a < LM; ++a)
                    au = fma(u[a] - ug[a], s_RR[a][i], au);
                o.lu[(b * T + t) * LM + i] = au;
            }
        } else if (t == T) { // terminal l_x, l_xx (ilqr_core.py:120-121), Q_terminal read from memory
            const double* xg = P.xg + b * P.gsx + (PATH ? (int64_t)T * LN : 0);
            double acc = (s_x[s][0] - xg[0]) * (P.Qt[0 * LN + i] + P.Qt[i * LN + 0]);
            for (int a = 1; a < LN; ++a)
                acc = fma(s_x[s][a] - xg[a], P.Qt[a * LN + i] + P.Qt[i * LN + a], acc);
            o.lxT[b * LN + i] = acc;
            for (int a = 0; a < LN; ++a)
                o.lxxT[(b * LN + a) * LN + i] = P.Qt[a * LN + i] + P.Qt[i * LN + a];
        }
    }
    __syncthreads();
    const int ns = (T - t0) < DS ? (T - t0) : DS; // steps t < T of this chunk
    for (int s = 0; s < ns; ++s) {
        const int64_t bt = b * T + t0 + s;
        // 16 bytes per lane and store: elements e, e + 1 of a row (same i, columns j and j + 1)
        for (int e = 2 * tid; e < LN * LN; e += 512) {
            int i = e >> 5, j = e & 31;
            double v0 = (i == j ? 1.0 : 0.0) + P.dt * (P.A[e] + s_g[s][i] * s_W[i][j]);
            double v1 = (i == j + 1 ? 1.0 : 0.0) + P.dt * (P.A[e + 1] + s_g[s][i] * s_W[i][j + 1]);
            *reinterpret_cast<double2*>(o.fx + bt * LN * LN + e) = make_double2(v0, v1);
            *reinterpret_cast<double2*>(o.lxx + bt * LN * LN + e) = make_double2(s_QQ[i][j], s_QQ[i][j + 1]);
        }
        if (tid < LN * LM / 2) {
            *reinterpret_cast<double2*>(o.fu + bt * LN * LM + 2 * tid) = make_double2(P.dt * P.B[2 * tid], P.dt * P.B[2 * tid + 1]);
            *reinterpret_cast<double2*>(o.lux + bt * LM * LN + 2 * tid) = make_double2(0.0, 0.0);
        }
        if (tid < LM * LM)
            o.luu[bt * LM * LM + tid] = s_RR[tid / LM][tid % LM];
    }
}

// ------------------------------------------------------------------ host side

int large_check(const ilqr_large_model* m)
{
    if (!m)
        return ILQR_ERR_ARG;
#ifdef ILQR_LARGE_USER_MODEL
    if (m->n != UN || m->m != UM) // the library was compiled around a model of these dimensions
        return ILQR_ERR_UNSUPPORTED;
    if (m->T < 1 || !m->A || !m->Q || !m->R || !m->Q_terminal || !m->x_goal || !m->u_goal || !m->lower || !m->upper)
        return ILQR_ERR_ARG; // (A: the model's parameter vector; B and W are not used)
#else
    if (m->n < 1 || m->n > LN || m->m < 1 || m->m > LM)
        return ILQR_ERR_UNSUPPORTED;
    if (m->T < 1 || !m->A || !m->B || !m->W || !m->Q || !m->R || !m->Q_terminal || !m->x_goal || !m->u_goal ||
        !m->lower || !m->upper)
        return ILQR_ERR_ARG;
#endif
    if (m->cost_kind != ILQR_COST_QR && m->cost_kind != ILQR_COST_PATH_QR)
        return ILQR_ERR_UNSUPPORTED;
    if ((m->goal_stride_x != 0) != (m->goal_stride_u != 0))
        return ILQR_ERR_ARG;
    if (m->goal_stride_x != 0) { // one goal or path per problem
        const bool path = m->cost_kind == ILQR_COST_PATH_QR;
        if (m->goal_stride_x < (path ? ((int64_t)m->T + 1) * LN : LN) || m->goal_stride_u < (path ? (int64_t)m->T * LM : LM))
            return ILQR_ERR_ARG;
    }
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
    P.path = m->cost_kind == ILQR_COST_PATH_QR ? 1 : 0;
    P.gsx = m->goal_stride_x;
    P.gsu = m->goal_stride_u;
    return P;
}

int large_qp(const ilqr_opts* o, LQPOpts& q, const ilqr_large_model* m)
{
    q.m_real = m->m;
    q.real_mask = (1u << m->m) - 1u;
    if (!o || o->qp_max_iter < 0 || !(o->qp_step_dec > 0.0) || !(o->qp_step_dec < 1.0) || !(o->qp_min_step > 0.0))
        return ILQR_ERR_ARG;
    q.max_iter = o->qp_max_iter;
    q.min_grad = o->qp_min_grad;
    q.min_rel = o->qp_min_rel_improve;
    q.step_dec = o->qp_step_dec;
    q.min_step = o->qp_min_step;
    q.armijo = o->qp_armijo;
    volatile double step = 1.0;
    int n = 0;
    do {
        step = step * o->qp_step_dec;
        ++n;
    } while (!(step < o->qp_min_step) && n < 1000000);
    q.n_exhaust = n;
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
    if (model->cost_kind == ILQR_COST_PATH_QR) {
        hipLaunchKernelGGL(k_large_rollout_cost<true>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, ls(stream), large_fill(model), B, x0, us,
                           xs, J);
    } else {
        hipLaunchKernelGGL(k_large_rollout_cost<false>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, ls(stream), large_fill(model), B, x0, us,
                           xs, J);
    }
    return lrc(hipGetLastError());
}

int ilqr_large_deriv(const ilqr_large_model* model, int64_t B, const double* xs, const double* us,
                     const ilqr_deriv_bm_t* out, void* stream)
{
    int rc = large_check(model);
    if (rc != ILQR_OK)
        return rc;
#ifdef ILQR_LARGE_USER_MODEL
    return ILQR_ERR_UNSUPPORTED; // the derivative stream is the built-in model's
#endif
    if (B < 1 || !xs || !us || !out || !out->fx || !out->fu || !out->lx || !out->lu || !out->lxx || !out->luu ||
        !out->lux || !out->lxT || !out->lxxT)
        return ILQR_ERR_ARG;
    const int64_t chunks = (model->T + 1 + DS - 1) / DS;
    if (B * chunks > 0x7fffffffLL)
        return ILQR_ERR_ARG;
    dim3 grid((unsigned)(B * chunks));
    if (model->cost_kind == ILQR_COST_PATH_QR)
        hipLaunchKernelGGL(k_large_deriv<true>, grid, dim3(256), 0, ls(stream), large_fill(model), B, xs, us, *out);
    else
        hipLaunchKernelGGL(k_large_deriv<false>, grid, dim3(256), 0, ls(stream), large_fill(model), B, xs, us, *out);
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
    rc = large_qp(opts, q, model);
    if (rc != ILQR_OK)
        return rc;
    if (B < 1 || !xs || !us || !mu || !k || !K || !dJ || !info)
        return ILQR_ERR_ARG;
    const bool path = model->cost_kind == ILQR_COST_PATH_QR, full = model->m == LM;
    const dim3 grid((unsigned)((B + 3) / 4));
#define ILQR_LAUNCH_LB(PATH_, FULL_)                                                                                   \
    hipLaunchKernelGGL((k_large_backward<PATH_, FULL_>), grid, dim3(256), 0, ls(stream), large_fill(model), q, B, xs, us, \
                       mu, k, K, dJ, info, qp_result, free_mask)
    if (path && full)
        ILQR_LAUNCH_LB(true, true);
    else if (path)
        ILQR_LAUNCH_LB(true, false);
    else if (full)
        ILQR_LAUNCH_LB(false, true);
    else
        ILQR_LAUNCH_LB(false, false);
#undef ILQR_LAUNCH_LB
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
    if (model->cost_kind == ILQR_COST_PATH_QR) {
        hipLaunchKernelGGL(k_large_forward_ls<true>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, ls(stream), large_fill(model), al,
                           (int)n_alpha, (int)store_idx, B, xs, us, k, K, J_new, xs_new, us_new);
    } else {
        hipLaunchKernelGGL(k_large_forward_ls<false>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, ls(stream), large_fill(model), al,
                           (int)n_alpha, (int)store_idx, B, xs, us, k, K, J_new, xs_new, us_new);
    }
    return lrc(hipGetLastError());
}

#ifdef ILQR_LARGE_QP_PROF
int ilqr_large_debug_qp_phases(unsigned long long* host8, int reset)
{
    unsigned long long z[8] = {0};
    if (reset)
        return lrc(hipMemcpyToSymbol(HIP_SYMBOL(g_lqp), z, sizeof(z)));
    return lrc(hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_lqp), sizeof(z)));
}
#endif

#ifdef ILQR_LARGE_USER_MODEL
/* The dimensions the library was compiled for (states, controls, parameters). */
int ilqr_large_user_dims(int32_t* n, int32_t* m, int32_t* n_params)
{
    if (!n || !m || !n_params)
        return ILQR_ERR_ARG;
    *n = UN;
    *m = UM;
    *n_params = ILQR_USER_NP;
    return ILQR_OK;
}
#endif

int ilqr_large_debug_profile(void* device_buffer)
{
    unsigned long long* p = static_cast<unsigned long long*>(device_buffer);
    return lrc(hipMemcpyToSymbol(HIP_SYMBOL(g_large_prof), &p, sizeof(p)));
}

size_t ilqr_large_workspace_bytes(const ilqr_large_model* model, int64_t B)
{
    if (large_check(model) != ILQR_OK || B < 1)
        return 0;
    size_t T = (size_t)model->T;
    // second copies of xs, us, k, K and (256 bytes, at the end) the problem queue's counter
    return (size_t)B * sizeof(double) * ((T + 1) * LN + T * LM + T * LM + T * LM * LN) + 256;
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
    O.dJ_carry = opts->dJ_carry;
    for (int i = 0; i < 32; ++i)
        O.alphas[i] = i < opts->n_alpha ? opts->alphas[i] : 0.0;
    rc = large_qp(opts, O.qp, model);
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
    p += (size_t)B * T * LM * LN;
    int* queue = reinterpret_cast<int*>(p);
    if (B > 0x7fffffff - 4096)
        return ILQR_ERR_ARG; // the queue's counter is a 32-bit integer
    hipError_t e = hipMemsetAsync(queue, 0, sizeof(int), ls(stream));
    if (e != hipSuccess)
        return lrc(e);
    // one workgroup (4 wavefronts, 123 KB of LDS) per CU is resident at a time; more would only start to
    // find the queue empty
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1)
        n_cu = 256;
    const int64_t groups = (B + 3) / 4 < n_cu ? (B + 3) / 4 : n_cu;
    ilqr_trace tr;
    if (trace)
        tr = *trace;
    else
        std::memset(&tr, 0, sizeof(tr));
    const bool path = model->cost_kind == ILQR_COST_PATH_QR, full = model->m == LM;
#define ILQR_LAUNCH_LS(PATH_, FULL_)                                                                                   \
    hipLaunchKernelGGL((k_large_solve<PATH_, FULL_>), dim3((unsigned)groups), dim3(256), 0, ls(stream), large_fill(model), \
                       O, B, x0, us_init, delta, k, K, xs, us, W, queue, J, mu, iters, status, tr)
    if (path && full)
        ILQR_LAUNCH_LS(true, true);
    else if (path)
        ILQR_LAUNCH_LS(true, false);
    else if (full)
        ILQR_LAUNCH_LS(false, true);
    else
        ILQR_LAUNCH_LS(false, false);
#undef ILQR_LAUNCH_LS
    return lrc(hipGetLastError());
}

} // extern "C"
