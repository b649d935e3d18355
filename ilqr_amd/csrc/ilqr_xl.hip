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

// All of a group of loaded values in registers at this point: the compiler, short of registers in these kernels, would
// otherwise issue a load, wait for it, use it, and only then issue the next (one LDS latency per term of a chain).
#define XL_PIN4(a) asm volatile("" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]))
#define XL_PIN16(a)                                                                                                 \
    asm volatile(""                                                                                                  \
                 : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]), "+v"((a)[4]), "+v"((a)[5]), "+v"((a)[6]),  \
                   "+v"((a)[7]), "+v"((a)[8]), "+v"((a)[9]), "+v"((a)[10]), "+v"((a)[11]), "+v"((a)[12]),              \
                   "+v"((a)[13]), "+v"((a)[14]), "+v"((a)[15]))

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
    int n_exhaust; // number of step reductions after which step < min_step (host-simulated)
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

constexpr int XTAB = 512; // Armijo step sizes tabulated per sweep (the loop's own repeated multiplications)

// A backward sweep.  INVARIANT: every matrix below holds exact zeros outside its rows x cols (the sweep starts by
// zeroing everything and every store is guarded), so the MFMA operands are read without bounds checks: what lies
// outside a matrix appends fma(0, 0, s) == s to the end of a chain (s may be anything, NaN included).
struct XlBw {
    double Vxx[XN * LDN], fx[XN * LDN];
    double M1[XN * LDN]; // f_x^T V_xx; later in the step the unsymmetrised V_xx
    double fu[XN * LDM];
    double M2[XM * LDN];  // f_u^T (V_xx + mu I)
    double T1[XN * LDM];  // K^T Q_uu
    double Qux[XM * LDN], Kl[XM * LDN];
    double H[XM * LDM];  // Q_uu
    double Lf[XM * XM];  // the Cholesky factor the triangular solves read (zeros in the rows / columns of clamped variables)
    double steps[XTAB];
    // operands of a step, written one step ahead by wavefronts 1..3 while wavefront 0 runs boxQP: two sets, used in
    // turn by the parity of the step
    double xd[2][XN], lx[2][XN];
    double ud[2][XM], lu[2][XM], lo[2][XM], hi[2][XM];
    double xw[3][XN], gi[3][XN]; // per wavefront 1..3: its own copy of x and of c phi'(W x)
    double Vx[XN], Qx[XN];
    double Qu[XM], qx[XM], grad[XM];
};

struct XlFwWave { // a rollout of one wavefront
    double xn[XN], dv[XN], dg[2][XN], acc[XN], un[XM], ud[XM], acu[XM];
    double Kst[XM * LDN]; // K_t of the step in hand
};

struct XlFw { // the rollouts of a line search: the model transposed (a lane per row reads consecutive words)
    double At[XN * LDN], Wt[XN * LDN], Qr[XN * LDN], Bt[XM * LDN], Rr[XM * LDM];
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
    int arrived;       // wavefronts 1..3 that have written their rows of M1, counted up through a sweep
};

static_assert(sizeof(XlMem) <= 160 * 1024, "one problem's matrices must fit the LDS of a CU");

#ifdef ILQR_XL_PROF // diagnostic build only: shader clocks of wavefront 0 per phase of a backward step, summed
static __device__ unsigned long long g_xl_prof[16];
#define XP_T(var) const long long var = clock64()
#define XP_ADD(slot, a, b)                                                       \
    if ((threadIdx.x & 255) == 0)                                                \
    atomicAdd(&g_xl_prof[slot], (unsigned long long)((b) - (a)))
#else
#define XP_T(var)
#define XP_ADD(slot, a, b)
#endif

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

// Tile (it, jt) of A B', the chain of every element running over kb k-blocks of four (k = 0 ... 4 kb - 1; what lies
// beyond the matrices reads as zero, see XlBw).  TA: A is stored transposed (element (i, k) at A[k * lda + i]), else
// at A[i * lda + k]; TB likewise for element (k, j) of B.  DIAG: B' = B + diag I for k < n (V_xx + mu * eye,
// ilqr_core.py:210).  The operands of four k-blocks are loaded together, ahead of their MFMAs.
template <bool TA, bool TB, bool DIAG = false>
__device__ __forceinline__ d4 xl_tile(const double* A, int lda, const double* Bm, int ldb, int it, int jt, int kb, int g,
                                      int c, double diag = 0.0, int n = 0)
{
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    const int i = 16 * it + c, j = 16 * jt + c;
    const double* pa = TA ? A + g * lda + i : A + i * lda + g;
    const double* pb = TB ? Bm + j * ldb + g : Bm + g * ldb + j;
    const int sa = TA ? 4 * lda : 4, sb = TB ? 4 : 4 * ldb;
    int q = 0;
    for (; q + 4 <= kb; q += 4) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = pa[u * sa];
            b[u] = pb[u * sb];
        }
        XL_PIN4(a);
        XL_PIN4(b);
        if (DIAG) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * (q + u) + g;
                if (k < n)
                    b[u] = b[u] + (k == j ? diag : 0.0);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            acc = xl_mfma(a[u], b[u], acc);
        pa += 4 * sa;
        pb += 4 * sb;
    }
    for (; q < kb; ++q) {
        double a = pa[0], b = pb[0];
        if (DIAG) {
            const int k = 4 * q + g;
            if (k < n)
                b = b + (k == j ? diag : 0.0);
        }
        acc = xl_mfma(a, b, acc);
        pa += sa;
        pb += sb;
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
//
// boxQP and everything around it in a backward step is the work of wavefront 0 alone, and none of it reads memory
// inside a chain: lane a < 16 owns variable a and keeps row a of Q_uu and of its Cholesky factor in registers; what a
// chain needs from another row is a lane read with a constant lane; 16-vectors that every lane needs (the iterate,
// the search direction) are held replicated.  Variables beyond m do not exist for the reference: they are exact
// zeros outside every mask, and so are the rows and columns of clamped variables in the factor (the reference
// compacts the free variables, ilqr_boxQP.py:114-132; fma(a, 0, s) == s keeps the same terms in the same order).

// v of `lane` (wave-uniform index) on every lane.  The result is handed on as an ordinary vector register: the
// compiler would otherwise keep every replicated number in scalar registers, of which there are far too few for
// the replicated 16-vectors below (they ended up spilled, lane by lane, into vector registers and from there into
// scratch memory).
__device__ __forceinline__ double xl_rl(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    double r = __hiloint2double(hi, lo);
    asm volatile("" : "+v"(r)); // (volatile: not to be merged with an equal read elsewhere and kept alive in between)
    return r;
}
// a condition that is the same on every lane, as a scalar (keeps the control flow around it wave-uniform)
__device__ __forceinline__ unsigned long long xl_uni_u64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <class T>
__device__ __forceinline__ T* xl_uni_ptr(T* p)
{
    return reinterpret_cast<T*>(xl_uni_u64(reinterpret_cast<unsigned long long>(p)));
}

__device__ __forceinline__ bool xl_uni(bool c) { return __builtin_amdgcn_readfirstlane(c ? 1 : 0) != 0; }

struct XlFac {
    double Lr[XM]; // lane a: L[a][j] (0 for j > a and in the rows / columns of clamped variables)
    double rd[XM]; // 1 / L[j][j], the same on every lane (1 for a clamped j)
    unsigned fm;   // the free set it belongs to
};

// Lower Cholesky factor of the rows/columns `fm` of H (Hr: lane a holds row a): column by column the pivot's chain,
// square root and reciprocal, then every row below it -- each entry the k-ascending chain
// t = fma(-L[a][k], L[j][k], t) of LAPACK's dpotf2 over the free k, scaled by the reciprocal.  Lane a keeps its row
// in registers and writes every finished entry to Lm (LDS, [16][16]), from where the pivot row of the next column is
// read back with broadcast loads (all at once, ahead of the chain) and where the triangular solves find the factor.
// False on a non-positive (or NaN) pivot.
__device__ __forceinline__ bool xl_chol(const double (&Hr)[XM], unsigned fm, int lane, XlFac& F, double* Lm)
{
    const bool fa = lane < XM && ((fm >> lane) & 1u);
    const int la = lane & (XM - 1);
    F.fm = fm;
#pragma unroll
    for (int j = 0; j < XM; ++j) {
        if (!((fm >> j) & 1u)) {
            F.Lr[j] = 0.0;
            F.rd[j] = 1.0;
            if (lane < XM)
                Lm[la * XM + j] = 0.0;
            continue;
        }
        double lj[XM]; // row j of the factor, columns < j
#pragma unroll
        for (int k = 0; k < XM; ++k)
            lj[k] = k < j ? Lm[j * XM + k] : 0.0;
        XL_PIN16(lj);
        double t = Hr[j];
#pragma unroll
        for (int k = 0; k < j; ++k)
            t = fma(-F.Lr[k], lj[k], t);
        const double s = xl_rl(t, j);
        if (!xl_uni(s > 0.0))
            return false;
        const double d = sqrt(s);
        const double r = 1.0 / d;
        F.Lr[j] = lane == j ? d : ((fa && lane > j) ? t * r : 0.0);
        F.rd[j] = r;
        if (lane < XM)
            Lm[la * XM + j] = F.Lr[j];
        xl_wave_sync();
    }
    return true;
}

// (L L^T) x = b in place for the rows/columns fm (cho_solve, each substitution multiplying by the reciprocal of
// the diagonal entry); every lane solves its own right-hand side (all the same one, or one per lane).  Entries of
// clamped variables must be 0 on entry and are 0 on exit.  Lm: the factor in LDS; a row's entries
// are loaded while the chain of the row before runs.  (xl_chol leaves the factor there.)
// ONES: the factor is the all-ones matrix boxQP starts with (H_ff = np.ones((n, n)), ilqr_boxQP.py:36), fm = all.
template <bool ONES>
__device__ __forceinline__ void xl_chol_solve_t(const XlFac& F, const double* Lm, unsigned fm, double (&b)[XM])
{
    double nxt[XM];
#pragma unroll
    for (int j = 0; j < XM; ++j)
        nxt[j] = 0.0;
    if (!ONES)
        nxt[0] = Lm[1 * XM + 0];
#pragma unroll
    for (int i = 0; i < XM; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        double cur[XM];
#pragma unroll
        for (int j = 0; j < XM; ++j)
            cur[j] = nxt[j];
        if (!ONES && i + 1 < XM) {
#pragma unroll
            for (int j = 0; j <= i; ++j)
                nxt[j] = Lm[(i + 1) * XM + j];
        }
        double s = b[i];
#pragma unroll
        for (int j = 0; j < i; ++j)
            s = ONES ? fma(-1.0, b[j], s) : fma(-cur[j], b[j], s);
        b[i] = ((fm >> i) & 1u) ? (ONES ? s * 1.0 : s * F.rd[i]) : 0.0;
    }
    if (!ONES) { // column XM - 2 below the diagonal
        nxt[XM - 1] = Lm[(XM - 1) * XM + XM - 2];
    }
#pragma unroll
    for (int i = XM - 1; i >= 0; --i) {
        __builtin_amdgcn_sched_barrier(0);
        double cur[XM];
#pragma unroll
        for (int j = 0; j < XM; ++j)
            cur[j] = nxt[j];
        if (!ONES && i - 1 >= 0 && i - 1 < XM - 2) { // column i - 1: L[j][i - 1], j > i - 1 (column XM - 2 is in hand)
#pragma unroll
            for (int j = i; j < XM; ++j)
                nxt[j] = Lm[j * XM + (i - 1)];
        }
        double s = b[i];
#pragma unroll
        for (int j = i + 1; j < XM; ++j)
            s = ONES ? fma(-1.0, b[j], s) : fma(-cur[j], b[j], s);
        b[i] = ((fm >> i) & 1u) ? (ONES ? s * 1.0 : s * F.rd[i]) : 0.0;
    }
}
__device__ __forceinline__ void xl_chol_solve(const XlFac& F, const double* Lm, unsigned fm, bool ones, double (&b)[XM])
{
    if (ones)
        xl_chol_solve_t<true>(F, Lm, fm, b);
    else
        xl_chol_solve_t<false>(F, Lm, fm, b);
}

// ---- wavefront 0 of a backward step: PD test, warm start, boxQP, gains ------------------------------------------

struct XlQpOut {
    int flag;           // XL_STEP_*
    int n_iter, n_backtrack;
    bool clamped;       // a control is clamped at the solution
    double dJ1, dJ2;    // k.Q_u and 0.5 k.(Q_uu k)
};

enum { XL_STEP_OK = 0, XL_STEP_NOT_PD = 1, XL_STEP_LINALG = 2 };

// objective x.g + 0.5 x^T H x of a candidate held in registers (ilqr_boxQP.py:62, :148), every chain in the
// reference's order: (x^T H)_j for two columns j at a time -- a loop that is NOT unrolled (see below) -- then the
// chain over the columns, for which the candidate is also kept in LDS (xq [16][64], the lane's own column; a
// register array cannot be indexed by the loop counter).  g = W.Qu and H = W.H are read from LDS (zeros beyond m).
__device__ __forceinline__ double xl_value(const XlBw& W, double* xq, int lane, const double (&x)[XM])
{
#pragma unroll
    for (int i = 0; i < XM; ++i)
        xq[i * 64 + lane] = x[i];
    double xg = x[0] * W.Qu[0];
#pragma unroll
    for (int i = 1; i < XM; ++i)
        xg = fma(x[i], W.Qu[i], xg);
    double tot = 0.0;
#pragma unroll 1
    for (int j = 0; j < XM; j += 2) {
        const double* h = W.H + j;
        double h0[XM], h1[XM];
#pragma unroll
        for (int i = 0; i < XM; ++i) {
            h0[i] = h[i * LDM];
            h1[i] = h[i * LDM + 1];
        }
        XL_PIN16(h0);
        XL_PIN16(h1);
        double a0 = x[0] * h0[0], a1 = x[0] * h1[0];
#pragma unroll
        for (int i = 1; i < XM; ++i) {
            a0 = fma(x[i], h0[i], a0);
            a1 = fma(x[i], h1[i], a1);
        }
        const double x0 = xq[j * 64 + lane], x1 = xq[(j + 1) * 64 + lane];
        tot = (j == 0) ? a0 * x0 : fma(a0, x0, tot);
        tot = fma(a1, x1, tot);
    }
    return xg + 0.5 * tot;
}

// Armijo loop, proof of exhaustion.  Near a solution the reference's backtracking loop (ilqr_boxQP.py:151-159) often
// runs until the step falls below min_step: the search direction is tiny, every candidate's objective equals the old
// one up to rounding, and (vc - old) / (sdotg + 1e-10) stays below the Armijo constant because of the 1e-10.  Before
// 64 more literal trips are evaluated the wavefront tries to PROVE that every trip from step size `st` on continues:
//   |xc_i - x_i| <= E_i = (1 + 1e-7) st |s_i| + 1.2e-16 |x_i|   (two roundings; clamping only shortens it)
//   exact objective v:  |v(xc) - v(x)| <= sum_i E_i G_i + 1/2 sum_ij E_i |H_ij| E_j,   with
//       G_i = 1.01 (|grad_i| + 2.2e-15 (|g_i| + sum_j |H_ij| |x_j|) + 1/2 sum_j |H_ij - H_ji| |x_j|)
//       (computed gradient + the rounding error of its 16-term chain + the asymmetry of the computed Q_uu)
//   computed objective (a 16-term chain, 16 chains of 16 terms, a chain over their results, one add):
//       |V(z) - v(z)| <= 6e-15 A,   A = sum_i X_i |g_i| + sum_ij X_i |H_ij| X_j,  X_i = |x_i| + E_i >= |z_i|
//   hence |vc - old| <= NUM = 1.001 (sum E G + 1/2 E|H|E + 12e-15 A) + 1e-200, and the computed slope satisfies
//   |sdotg| <= S = 1.001 sum_i E_i |grad_i|, so the denominator is >= 0.999 (1e-10 - S).
// If S < 1e-10 and NUM < armijo * 0.999 (1e-10 - S), the quotient is below armijo at this step size, and -- all
// bounds grow with st -- at every smaller one: the loop ends at trip n_exhaust with result 2 and that trip's
// candidate.  Any NaN or infinity makes a comparison false (no proof, literal evaluation).  The bounds are generous
// by the factors shown.  Lane a < 16 forms the terms of variable a (Hr: its row of Q_uu, gl and grad: its entries of
// g and of the computed gradient); x and the search direction are replicated.
__device__ __forceinline__ bool xl_armijo_exhausts(const XlBw& W, const double (&Hr)[XM], int lane, double gl,
                                                   double grad, const double (&x)[XM], const double (&srch)[XM],
                                                   double st, double armijo)
{
    const int la = lane & (XM - 1);
    double hax = 0.0, asy = 0.0, hE = 0.0, hX = 0.0, Ea = 0.0, Xa = 0.0;
#pragma unroll
    for (int j = 0; j < XM; ++j) {
        const double axj = fabs(x[j]);
        const double Ej = 1.0000001 * st * fabs(srch[j]) + 1.2e-16 * axj;
        const double Xj = axj + Ej;
        const double h = fabs(Hr[j]);
        hax = fma(h, axj, hax);
        hE = fma(h, Ej, hE);
        hX = fma(h, Xj, hX);
        asy = fma(fabs(Hr[j] - W.H[j * LDM + la]), axj, asy);
        Ea = lane == j ? Ej : Ea;
        Xa = lane == j ? Xj : Xa;
    }
    const bool own = lane < XM;
    const double ag = fabs(gl), agr = fabs(grad);
    const double G = 1.01 * (agr + 2.2e-15 * (ag + hax) + 0.5 * asy);
    const double t1 = own ? Ea * G + 0.5 * (Ea * hE) : 0.0;
    const double t2 = own ? Xa * ag + Xa * hX : 0.0;
    const double t3 = own ? Ea * agr : 0.0;
    double d1 = 0.0, A = 0.0, Ss = 0.0;
#pragma unroll
    for (int a = 0; a < XM; ++a) {
        d1 += xl_rl(t1, a);
        A += xl_rl(t2, a);
        Ss += xl_rl(t3, a);
    }
    const double NUM = 1.001 * (d1 + 12e-15 * A) + 1e-200;
    const double S = 1.001 * Ss;
    return xl_uni((S < 1e-10) && (armijo > 0.0) && (NUM < armijo * (0.999 * (1e-10 - S))));
}

// np.all(np.linalg.eigvals(Q_uu) > 0) else break; R_Q_uu = cholesky(Q_uu); k = cho_solve(R_Q_uu, Q_u) (the warm
// start, positive sign as the reference has it); boxQP(Q_uu, Q_u, lower - u, upper - u, k) (ilqr_boxQP.py:9-186);
// K[free] = -cho_solve(H_ff, Q_ux[free]); Delta_J   (ilqr_core.py:134-176).
//
// ONE loop runs all of it, as a small state machine (warm start -> boxQP passes -> gains), so that the two long
// unrolled routines -- the factorisation and the triangular solves, ~10 and ~15 KB of instructions -- exist once
// each.  Written out at each of their five uses, the wavefront's code outgrew the instruction cache and every step
// fetched it from memory again (~40 cycles per instruction).
//
// In: W.H (Q_uu), W.Qu, W.lo[p], W.hi[p], W.Qux.  Out: k_row [m], K_rows [m][n] (global), W.qx (k), W.Kl (K), the
// optional records, and the return value.  Uses W.T1 as scratch and leaves it zero.
// Lane a < m owns variable a for the gradient, the classification and the right-hand side; the Armijo loop
// (:146-156) evaluates 64 consecutive trips at a time, one per lane, and takes the first that ends the loop.
__device__ __forceinline__ XlQpOut xl_qp_phase(XlBw& W, int p, const XlQP& O, int n, int m, int lane,
                                               double* __restrict__ k_row, double* __restrict__ K_rows,
                                               int32_t* free_row, int32_t* result_at)
{
    XlQpOut out{XL_STEP_OK, 0, 0, false, 0.0, 0.0};
    double Hr[XM]; // lane a < 16: row a of Q_uu
#pragma unroll
    for (int j = 0; j < XM; ++j)
        Hr[j] = lane < XM ? W.H[(lane & (XM - 1)) * LDM + j] : 0.0;
    {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < XM; ++j)
            bad = bad || !(fabs(Hr[j]) < __builtin_inf());
        if (__any(bad)) {
            out.flag = XL_STEP_LINALG; // eigvals raises LinAlgError on NaN / Inf
            return out;
        }
    }
    const unsigned all = (1u << m) - 1u;
    const int la = lane < XM ? lane : 0;
    const double* lov = W.lo[p];
    const double* hiv = W.hi[p];
    const double lo = lov[la], hi = hiv[la], gl = W.Qu[la]; // (zeros beyond m)
    double* xq = W.T1;

    enum { WARM, PASS, GAINS };
    int mode = WARM;
    XlFac F;
    F.fm = ~0u;
    double x[XM], srch[XM];
    unsigned cm = 0u, fm = 0u; // clamped set (none before the first classification, :28) and free set of boxQP
    bool factored = false;     // false: H_ff is still the all-ones matrix of :36
    int result = -1, iter = 0;
    double value = 0.0, oldvalue = 0.0, grad = 0.0;
    for (;;) {
        bool want_chol = false;
        unsigned fq = all;
        if (mode == WARM) {
            want_chol = true;
        } else if (mode == PASS) {
            if (iter >= O.max_iter || result != -1) {
                mode = GAINS;
            } else if (iter > 1 && xl_uni((oldvalue - value) < O.min_rel * fabs(oldvalue) && (oldvalue - value) > 0)) {
                result = 4;
                mode = GAINS;
            } else {
                oldvalue = value;
                // grad = g + H x (:82); clamped = at a bound with the gradient pointing outwards (:85-88)
                double acc = Hr[0] * x[0];
#pragma unroll
                for (int j = 1; j < XM; ++j)
                    acc = fma(Hr[j], x[j], acc);
                grad = gl + acc;
                double xa = 0.0; // the lane's own variable
#pragma unroll
                for (int a = 0; a < XM; ++a)
                    xa = lane == a ? x[a] : xa;
                const bool cl = lane < m && (((xa == lo) && (grad > 0)) || ((xa == hi) && (grad < 0)));
                const unsigned oldcm = cm;
                cm = (unsigned)__ballot(cl) & all;
                fm = all & ~cm;
                if (lane < XM)
                    W.grad[lane] = lane < m ? grad : 0.0;
                xl_wave_sync();
                if (cm == all) {
                    result = 6;
                    mode = GAINS;
                } else if (iter == 1 || oldcm != cm) { // :107-116
                    factored = true;
                    want_chol = fm != F.fm; // (else the same inputs would give the same bits)
                    fq = fm;
                }
            }
        }
        if (want_chol) {
            XP_T(f0);
            if (!xl_chol(Hr, fq, lane, F, W.Lf)) {
                out.flag = mode == WARM ? XL_STEP_NOT_PD : XL_STEP_LINALG;
                break;
            }
            XP_T(f1);
            XP_ADD(6, f0, f1);
        }
        if (mode == PASS) {
            // sla.norm(grad[free]) (:120); dnrm2 of a single element is its magnitude
            double ss = 0.0; // (fma(g, g, 0) is the product g * g: the chain starts at the first free variable)
#pragma unroll
            for (int a = 0; a < XM; ++a) {
                const double ga = ((fm >> a) & 1u) ? W.grad[a] : 0.0;
                ss = fma(ga, ga, ss);
            }
            const double gnorm = __popc(fm) == 1 ? fabs(W.grad[__ffs((int)fm) - 1]) : sqrt(ss);
            if (xl_uni(gnorm < O.min_grad)) {
                result = 5;
                mode = GAINS;
            } else {
                out.n_iter++;
            }
        }
        // the right-hand side: Q_u (warm start); g[free] + H_fc x[clamped] (:126-131; lane a forms entry a, the chain
        // over the clamped variables starts at the first of them: fma(h, x, 0) is the product); a column of Q_ux
        double b[XM];
        unsigned sm = all;
        bool ones = false;
        if (mode == WARM) {
#pragma unroll
            for (int a = 0; a < XM; ++a)
                b[a] = W.Qu[a];
        } else if (mode == PASS) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < XM; ++q)
                acc = fma(Hr[q], ((cm >> q) & 1u) ? x[q] : 0.0, acc);
            const double bv = (lane < XM && ((fm >> lane) & 1u)) ? gl + acc : 0.0;
#pragma unroll
            for (int a = 0; a < XM; ++a)
                b[a] = xl_rl(bv, a);
            sm = fm;
            ones = !factored;
        } else {
#pragma unroll
            for (int a = 0; a < XM; ++a)
                b[a] = ((fm >> a) & 1u) ? W.Qux[a * LDN + lane] : 0.0;
            sm = fm;
            ones = !factored;
        }
        XP_T(s0);
        if (sm != 0u)
            xl_chol_solve(F, W.Lf, sm, ones, b);
        XP_T(s1);
        XP_ADD(7, s0, s1);
        if (mode == WARM) {
#pragma unroll
            for (int a = 0; a < XM; ++a)
                x[a] = xl_clamp(lov[a], hiv[a], b[a]); // x = clamp(x0)   (:42)
            value = xl_value(W, xq, lane, x);
            mode = PASS;
            continue;
        }
        if (mode == GAINS) {
            // K[free, :] = -cho_solve(H_ff, Q_ux[free, :]), zero rows for the clamped controls (:157-160)
            if (lane < n) {
#pragma unroll
                for (int a = 0; a < XM; ++a)
                    if (a < m) {
                        const double v = ((fm >> a) & 1u) ? b[a] * -1 : 0.0;
                        W.Kl[a * LDN + lane] = v;
                        K_rows[(int64_t)a * n + lane] = v;
                    }
            }
            break;
        }
        // search[free] = -cho_solve(H_ff, .) - x[free]   (:132)
#pragma unroll
        for (int a = 0; a < XM; ++a)
            srch[a] = ((fm >> a) & 1u) ? b[a] * -1 - x[a] : 0.0;
        // Armijo loop (:141-156): trip t tries step_dec^t (formed by repeated multiplication, as the loop does: the
        // table holds those numbers, beyond it they are formed here); it ends at the first trip whose test fails
        // or, from the second trip on, whose step is below min_step.
        XP_T(a0);
        double base = 1.0;
        int t0 = 0;
        bool proven = false;
        for (;;) {
            if (proven) { // every trip from t0 on continues: the loop ends at trip n_exhaust with that trip's candidate
                const double se = W.steps[O.n_exhaust];
#pragma unroll
                for (int a = 0; a < XM; ++a)
                    x[a] = xl_clamp(lov[a], hiv[a], x[a] + se * srch[a]);
                value = xl_value(W, xq, lane, x);
                result = 2;
                out.n_backtrack += O.n_exhaust;
                break;
            }
            double mine;
            if (t0 + 64 <= XTAB) {
                mine = W.steps[t0 + lane];
                if (t0 + 64 == XTAB)
                    base = W.steps[XTAB - 1] * O.step_dec;
            } else {
                double st = base;
                mine = base;
                for (int q = 0; q < 64; ++q) {
                    if (q == lane)
                        mine = st;
                    st = st * O.step_dec;
                }
                base = st;
            }
            double xc[XM];
#pragma unroll
            for (int a = 0; a < XM; ++a)
                xc[a] = xl_clamp(lov[a], hiv[a], x[a] + mine * srch[a]);
            double sd = (xc[0] - x[0]) * W.grad[0];
#pragma unroll
            for (int a = 1; a < XM; ++a)
                sd = sd + (xc[a] - x[a]) * W.grad[a];
            const double vc = xl_value(W, xq, lane, xc);
            const bool cont = (vc - oldvalue) / (sd + 1e-10) < O.armijo;
            const bool exh = (t0 + lane >= 1) && (mine < O.min_step);
            const unsigned long long bal = __ballot(!cont || exh);
            if (bal != 0ull) {
                const int l = __ffsll((long long)bal) - 1;
#pragma unroll
                for (int a = 0; a < XM; ++a)
                    x[a] = xl_rl(xc[a], l);
                value = xl_rl(vc, l);
                if ((__ballot(exh) >> l) & 1ull)
                    result = 2;
                out.n_backtrack += t0 + l;
                break;
            }
            t0 += 64;
            // (the same proof for the trips from t0 on, before 64 more of them are evaluated)
            if (t0 <= O.n_exhaust && O.n_exhaust < XTAB)
                proven = xl_armijo_exhausts(W, Hr, lane, gl, grad, x, srch, W.steps[t0], O.armijo);
        }
        XP_T(a1);
        XP_ADD(8, a0, a1);
        xl_wave_sync(); // (W.grad is rewritten by the next pass)
        ++iter;
    }
#pragma unroll
    for (int a = 0; a < XM; ++a)
        xq[a * 64 + lane] = 0.0; // (T1 is zero outside its n x m)
    if (out.flag != XL_STEP_OK)
        return out;
    out.clamped = __popc(fm) < m;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < XM; ++a)
            W.qx[a] = x[a];
        if (result_at)
            *result_at = result;
    }
    xl_wave_sync();
    if (lane < m) {
        k_row[lane] = W.qx[lane];
        if (free_row)
            free_row[lane] = (int32_t)((fm >> lane) & 1u);
    }
    // Delta_J1 += k.Q_u ; Delta_J2 += 0.5 k.(Q_uu k)   (:175-176)
    double qa = Hr[0] * x[0];
#pragma unroll
    for (int q = 1; q < XM; ++q)
        qa = fma(Hr[q], x[q], qa);
    double d1 = x[0] * W.Qu[0];
#pragma unroll
    for (int q = 1; q < XM; ++q)
        d1 = fma(x[q], W.Qu[q], d1);
    double d2 = x[0] * xl_rl(qa, 0);
#pragma unroll
    for (int q = 1; q < XM; ++q)
        d2 = fma(x[q], xl_rl(qa, q), d2);
    out.dJ1 = d1;
    out.dJ2 = d2 * 0.5;
    return out;
}

// ---- backward sweep ---------------------------------------------------------------------------------------

struct XlBwOut {
    double dJ1, dJ2;
    int stopped_at, qp_iter, qp_backtrack, n_clamped, rc;
};

// What a backward step needs that does not depend on the recursion -- x_j, the cost gradients, c phi'(W x_j), and
// f_x = I + dt (A + diag(c phi') W) -- formed by wavefronts 1..3 (hw = 0..2) one step ahead, while wavefront 0 runs
// the boxQP of step j + 1.  Each wavefront forms c phi'(W x_j) for itself (no hand-over between them) and writes
// the column blocks jt = hw, hw + 3 of f_x: the blocks whose tiles of Q_xx it has just finished, so nobody still
// reads what it overwrites.  Wavefront 3 also forms l_x, wavefront 2 the control part (l_u and the box).
template <bool PATH>
__device__ __forceinline__ void xl_stage_step(const XlP& P, XlBw& W, const double* __restrict__ xs,
                                              const double* __restrict__ us, int j, int hw, int lane)
{
    const int n = __builtin_amdgcn_readfirstlane(P.n), m = __builtin_amdgcn_readfirstlane(P.m);
    const int NT = (n + 15) >> 4, p = j & 1, g = lane >> 4, c = lane & 15;
    const double* xg = PATH ? P.xg + (int64_t)j * n : P.xg;
    const double* ug = PATH ? P.ug + (int64_t)j * m : P.ug;
    if (lane < n) {
        const double xv = xs[(int64_t)j * n + lane];
        W.xw[hw][lane] = xv;
        if (hw == 2)
            W.xd[p][lane] = xv - xg[lane];
    }
    if (hw == 1 && lane < m) {
        const double uv = us[(int64_t)j * m + lane];
        W.ud[p][lane] = uv - ug[lane];
        W.lo[p][lane] = P.lower[lane] - uv;
        W.hi[p][lane] = P.upper[lane] - uv;
    }
    xl_wave_sync();
    if (lane < n) { // phi'(z) = (1 + z^2)^(-3/2), z = W x
        const double* wr = P.W + lane * n;
        double z = wr[0] * W.xw[hw][0];
        for (int q = 1; q < n; ++q)
            z = fma(wr[q], W.xw[hw][q], z);
        const double s = 1.0 + z * z;
        W.gi[hw][lane] = P.sig_c * (1.0 / (s * sqrt(s)));
    }
    if (hw == 2 && lane < n) { // l_x = x_diff^T (Q + Q^T)   (ilqr_cost.py:555-569)
        double acc = W.xd[p][0] * (P.Q[0 * n + lane] + P.Q[lane * n + 0]);
        for (int a = 1; a < n; ++a)
            acc = fma(W.xd[p][a], P.Q[a * n + lane] + P.Q[lane * n + a], acc);
        W.lx[p][lane] = acc;
    }
    if (hw == 1 && lane < m) { // l_u = u_diff^T (R + R^T)   (:571-587)
        double acc = W.ud[p][0] * (P.R[0 * m + lane] + P.R[lane * m + 0]);
        for (int a = 1; a < m; ++a)
            acc = fma(W.ud[p][a], P.R[a * m + lane] + P.R[lane * m + a], acc);
        W.lu[p][lane] = acc;
    }
    xl_wave_sync();
    for (int jt = hw; jt < NT; jt += 3) {
        const int b = 16 * jt + c;
        if (b < n)
            for (int a = g; a < n; a += 4)
                W.fx[a * LDN + b] = (a == b ? 1.0 : 0.0) + P.dt * (P.A[a * n + b] + W.gi[hw][a] * P.W[a * n + b]);
    }
}

// _backward_pass + _compute_Q_terms (ilqr_core.py:109-228) of one problem, run by the whole workgroup.
// k [T*m], K [T*m*n] are fully written (zeros where the sweep did not reach).
//
// A step: (1) all four wavefronts: M1 = f_x^T V_xx, M2 = f_u^T (V_xx + mu I), Q_x, Q_u; (2) Q_ux = M2 f_x and
// Q_uu = l_uu + M2 f_u; (3) wavefront 0: PD test, warm start, boxQP, gains -- the one serial part -- while
// wavefronts 1..3 form Q_xx = l_xx + M1 f_x (which boxQP does not need; each keeps the tiles of its column blocks in
// registers) and then stage the next step (xl_stage_step); (4) T1 = K^T Q_uu; (5) the value update, each tile by
// the wavefront that holds its Q_xx; (6) symmetrisation.
template <bool PATH>
__device__ __forceinline__ void xl_backward(const XlP& P, const XlQP& QO, XlMem& S, const double* __restrict__ xs,
                            const double* __restrict__ us, double mu, double* __restrict__ k, double* __restrict__ K,
                            XlBwOut& o, int32_t* qp_result, int32_t* free_mask)
{
    XlBw& W = S.bw;
    // (the sizes as scalars: every loop below is the same for all lanes)
    const int n = __builtin_amdgcn_readfirstlane(P.n), m = __builtin_amdgcn_readfirstlane(P.m),
              T = __builtin_amdgcn_readfirstlane(P.T);
    const int NT = (n + 15) >> 4, KN = (n + 3) >> 2, KM = (m + 3) >> 2;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lane0 = lane;
    if (qp_result)
        for (int i = tid; i < T; i += 256)
            qp_result[i] = -100; // step not reached
    if (free_mask)
        for (int i = tid; i < T * m; i += 256)
            free_mask[i] = 0;
    __syncthreads(); // (whoever used this memory before has finished)
    {
        double* z = reinterpret_cast<double*>(&W);
        for (int e = tid; e < (int)(sizeof(XlBw) / sizeof(double)); e += 256)
            z[e] = 0.0;
    }
    __syncthreads();
    // what stays the same for the whole sweep: f_u = dt B, the Armijo step table
    for (int e = tid; e < n * m; e += 256)
        W.fu[(e / m) * LDM + e % m] = P.dt * P.B[e];
    if (tid == 255) {
        double st = 1.0;
        for (int t = 0; t < XTAB; ++t) {
            W.steps[t] = st;
            st = st * QO.step_dec;
        }
    }
    // V_x = l_x(x_N), V_xx = l_xx(x_N)   (:127-128)
    if (tid < n)
        W.Qx[tid] = xs[(int64_t)T * n + tid] - (PATH ? P.xg[(int64_t)T * n + tid] : P.xg[tid]); // (x_diff, for a moment)
    __syncthreads();
    if (tid < n) {
        const int j = tid;
        double acc = W.Qx[0] * (P.Qt[0 * n + j] + P.Qt[j * n + 0]);
        for (int a = 1; a < n; ++a)
            acc = fma(W.Qx[a], P.Qt[a * n + j] + P.Qt[j * n + a], acc);
        W.Vx[j] = acc;
    }
    for (int e = tid; e < n * n; e += 256) {
        const int a = e / n, b = e % n;
        W.Vxx[a * LDN + b] = P.Qt[a * n + b] + P.Qt[b * n + a];
    }
    if (wave >= 1)
        xl_stage_step<PATH>(P, W, xs, us, T - 1, wave - 1, lane);
    if (tid == 0)
        S.arrived = 0;
    int n_arrivals = 0; // (steps this sweep has been through)
    double dJ1 = 0.0, dJ2 = 0.0; // (wave 0's; handed to the workgroup at the end)
    int qp_iter = 0, qp_bt = 0, n_cl = 0;
    o.stopped_at = -1;
    o.rc = 0;
    __syncthreads();

    const XlP& P0 = P;
    const double* const xs0 = xs;
    const double* const us0 = us;
    double* const k0 = k;
    double* const K0 = K;
    for (int i = T - 1; i >= 0; --i) {
        const int p = i & 1;
        // The arrays as this step sees them: their pointers are made opaque once per step, so that the addresses the
        // compiler derives from them are formed where they are used instead of being kept -- hundreds of registers'
        // worth -- across the steps.  (The model's pointers are left alone: with them in the list the backend could
        // not always keep them in scalar registers, and the sweep is no slower without.)
        XlP P = P0;
        const double* xs = xs0;
        const double* us = us0;
        double* k = k0;
        double* K = K0;
        asm volatile("" : "+s"(xs), "+s"(us), "+s"(k), "+s"(K));
        // (the same for what is derived from the lane index, the LDS addresses of every tile: afresh in every phase)
        int lane, g, c;
#define XL_FRESH_LANE()                 \
    lane = lane0;                       \
    asm volatile("" : "+v"(lane));      \
    g = lane >> 4;                      \
    c = lane & 15
        XL_FRESH_LANE();
        XP_T(c0);
        // -- (1) M2 = f_u^T (V_xx + mu I); Q_x, Q_u   (:200-212; M1 = f_x^T V_xx, which only Q_xx needs, is formed in
        // phase (3) beside the boxQP)
        for (int jt = wave; jt < NT; jt += 4)
            xl_tile_store(W.M2, LDN, m, n, 0, jt,
                          xl_tile<true, false, true>(W.fu, LDM, W.Vxx, LDN, 0, jt, KN, g, c, mu, n), g, c);
        // Q_x = l_x + f_x^T V_x, Q_u = l_u + f_u^T V_x (:206-207): lane = entry, the chain over the states 16 terms at a
        // time (operands beyond n are zeros: fma(0, 0, s) == s)
        if (wave == 3 || wave == 2) {
            const double* col = wave == 3 ? W.fx + lane : W.fu + (lane & (XM - 1));
            const int ld = wave == 3 ? LDN : LDM;
            double acc = 0.0;
            for (int a0 = 0; a0 < 16 * NT; a0 += 16) {
                double fv[16], vv[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    fv[q] = col[(a0 + q) * ld];
                    vv[q] = W.Vx[a0 + q];
                }
                XL_PIN16(fv);
                XL_PIN16(vv);
                acc = a0 == 0 ? fv[0] * vv[0] : fma(fv[0], vv[0], acc);
#pragma unroll
                for (int q = 1; q < 16; ++q)
                    acc = fma(fv[q], vv[q], acc);
            }
            if (wave == 3 && lane < n)
                W.Qx[lane] = W.lx[p][lane] + acc;
            if (wave == 2 && lane < m)
                W.Qu[lane] = W.lu[p][lane] + acc;
        }
        __syncthreads();
        XP_T(c1);
        XP_ADD(0, c0, c1);
        XL_FRESH_LANE();
        // -- (2) Q_ux = M2 f_x, Q_uu = l_uu + M2 f_u
        for (int q = wave; q < NT + 1; q += 4) {
            if (q < NT) {
                d4 acc = xl_tile<false, false>(W.M2, LDN, W.fx, LDN, 0, q, KN, g, c);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = 0.0 + acc[r]; // l_ux is identically zero
                xl_tile_store(W.Qux, LDN, m, n, 0, q, acc, g, c);
            } else {
                d4 acc = xl_tile<false, false>(W.M2, LDN, W.fu, LDM, 0, 0, KN, g, c);
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
        XP_T(c2);
        XP_ADD(1, c1, c2);
        XL_FRESH_LANE();
        // -- (3) wave 0: PD test, warm start, boxQP, gains, Delta_J (:134-176); waves 1..3: Q_xx, then the next step
        d4 qxx[8]; // tiles (it, jt = hw + 3 bs) at [4 bs + it]
        if (wave == 0) {
            const XlQpOut q = xl_qp_phase(W, p, QO, n, m, lane, k + (int64_t)i * m, K + (int64_t)i * m * n,
                                          free_mask ? free_mask + (int64_t)i * m : nullptr,
                                          qp_result ? qp_result + i : nullptr);
            if (q.flag == XL_STEP_OK) {
                qp_iter += q.n_iter;
                qp_bt += q.n_backtrack;
                if (q.clamped)
                    ++n_cl;
                dJ1 = dJ1 + q.dJ1;
                dJ2 = dJ2 + q.dJ2;
            }
            if (lane == 0)
                S.res_i[0] = q.flag;
            XP_T(c2c);
            XP_ADD(12, c2, c2c);
        } else {
            const int hw = wave - 1;
            // M1 = f_x^T V_xx (:208): this wavefront forms the row blocks hw and hw + 3 -- they read the column blocks
            // of f_x that are its own (it rewrites them for the next step further down) -- and Q_xx = l_xx + M1 f_x
            // below needs every row block: the three wavefronts meet at a counter in LDS in between (wavefront 0 is
            // busy with the boxQP and takes no part, so it is not a workgroup barrier).
            for (int bs = 0; bs < 2; ++bs) {
                const int it = hw + 3 * bs;
                if (it < NT)
                    for (int jt = 0; jt < NT; ++jt)
                        xl_tile_store(W.M1, LDN, n, n, it, jt,
                                      xl_tile<true, false>(W.fx, LDN, W.Vxx, LDN, it, jt, KN, g, c), g, c);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0)
                __hip_atomic_fetch_add(&S.arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&S.arrived, __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_WORKGROUP)) < 3 * (n_arrivals + 1))
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int bs = 0; bs < 2; ++bs)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int jt = hw + 3 * bs;
                    d4 v = d4{0.0, 0.0, 0.0, 0.0};
                    if (jt < NT && it < NT) {
                        const d4 acc = xl_tile<false, false>(W.M1, LDN, W.fx, LDN, it, jt, KN, g, c);
                        const int b = 16 * jt + c;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int a = 16 * it + 4 * r + g;
                            v[r] = (a < n && b < n) ? (P.Q[a * n + b] + P.Q[b * n + a]) + acc[r] : 0.0;
                        }
                    }
                    qxx[4 * bs + it] = v;
                }
            xl_wave_sync(); // (this wavefront's reads of its f_x blocks are done)
            XL_FRESH_LANE();
            if (i > 0)
                xl_stage_step<PATH>(P, W, xs, us, i - 1, hw, lane);
        }
        ++n_arrivals;
        __syncthreads();
        XP_T(c3);
        XP_ADD(2, c2, c3);
        XL_FRESH_LANE();
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
        // -- (4) T1 = K^T Q_uu
        for (int it = wave; it < NT; it += 4)
            xl_tile_store(W.T1, LDM, n, m, it, 0, xl_tile<true, false>(W.Kl, LDN, W.H, LDM, it, 0, KM, g, c), g, c);
        __syncthreads();
        XP_T(c4);
        XP_ADD(3, c3, c4);
        XL_FRESH_LANE();
        // -- (5) V_xx = Q_xx + (K^T Q_uu) K + K^T Q_ux + Q_ux^T K (unsymmetrised, into M1's place); V_x   (:164-167)
        if (wave == 0) {
            if (lane < n) {
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
        } else {
            const int hw = wave - 1;
#pragma unroll
            for (int bs = 0; bs < 2; ++bs)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int jt = hw + 3 * bs;
                    if (jt < NT && it < NT) {
                        // the three products of a tile, (K^T Q_uu) K, K^T Q_ux and Q_ux^T K: their operands -- four
                        // k-blocks each, rows beyond m are the zeros of the invariant -- are all requested before the
                        // first MFMA, so that the tile waits for LDS once instead of three times
                        d4 t1 = d4{0.0, 0.0, 0.0, 0.0}, t2 = t1, t3 = t1;
                        {
                            const int ii = 16 * it + c, jj = 16 * jt + c;
                            const double* pa1 = W.T1 + ii * LDM + g;
                            const double* pkj = W.Kl + g * LDN + jj;
                            const double* pki = W.Kl + g * LDN + ii;
                            const double* pqj = W.Qux + g * LDN + jj;
                            const double* pqi = W.Qux + g * LDN + ii;
                            double a1[4], kj[4], ki[4], qj[4], qi[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                a1[u] = pa1[4 * u];
                                kj[u] = pkj[4 * u * LDN];
                                ki[u] = pki[4 * u * LDN];
                                qj[u] = pqj[4 * u * LDN];
                                qi[u] = pqi[4 * u * LDN];
                            }
                            XL_PIN4(a1);
                            XL_PIN4(kj);
                            XL_PIN4(ki);
                            XL_PIN4(qj);
                            XL_PIN4(qi);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                t1 = xl_mfma(a1[u], kj[u], t1);
                                t2 = xl_mfma(ki[u], qj[u], t2);
                                t3 = xl_mfma(qi[u], kj[u], t3);
                            }
                        }
                        d4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[r] = ((qxx[4 * bs + it][r] + t1[r]) + t2[r]) + t3[r];
                        xl_tile_store(W.M1, LDN, n, n, it, jt, v, g, c);
                    }
                }
        }
        __syncthreads();
        XP_T(c5);
        XP_ADD(4, c4, c5);
        XL_FRESH_LANE();
        // -- (6) V_xx = 0.5 (V_xx + V_xx^T)   (:168)
        for (int a = wave; a < n; a += 4)
            if (lane < n)
                W.Vxx[a * LDN + lane] = 0.5 * (W.M1[a * LDN + lane] + W.M1[lane * LDN + a]);
        __syncthreads();
        XP_T(c6);
        XP_ADD(5, c5, c6);
#undef XL_FRESH_LANE
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
    for (int e = threadIdx.x; e < m * m; e += 256)
        F.Rr[(e / m) * LDM + e % m] = P.R[e];
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
//
// A step is two stretches of straight-line code with a wavefront barrier after each.  In the first, lane r forms the
// chains of row r of A x, W x and of column r of x_diff^T Q, and lane a < m the chain of row a of K (x_new - x) --
// four independent chains side by side, eight terms per round of LDS reads; in the second B u, the cost's chains
// over the columns and the new state.  What a step reads from memory -- the old state, control, k and column `lane`
// of K_t (which passes through LDS so that a lane can read its row), the goals of a path cost -- is fetched two to
// three steps ahead into registers.  Every chain has the reference's terms in the reference's order.
template <bool PATH>
__device__ double xl_forward(const XlP& P, XlMem& S, const double* __restrict__ xs, const double* __restrict__ us,
                             const double* __restrict__ k, const double* __restrict__ K, const double* __restrict__ x0,
                             double alpha, double* __restrict__ xs_new, double* __restrict__ us_new)
{
    // (the arguments are the same on all lanes of the wavefront, but derived from its index: as scalars, every
    // address below is a scalar base plus the lane's offset)
    xs = xl_uni_ptr(xs);
    us = xl_uni_ptr(us);
    k = xl_uni_ptr(k);
    K = xl_uni_ptr(K);
    x0 = xl_uni_ptr(x0);
    xs_new = xl_uni_ptr(xs_new);
    us_new = xl_uni_ptr(us_new);
    alpha = __longlong_as_double((long long)xl_uni_u64((unsigned long long)__double_as_longlong(alpha)));
    // Where the compiler keeps this function out of line, S and the arrays are generic pointers to it, and a generic
    // access is slower than an LDS or a global one and counts as both when waited for: S is said to be LDS, the
    // arrays are said to be global memory.
#if defined(__HIP_DEVICE_COMPILE__) // (the builtin exists in the device pass only)
    __builtin_assume(__builtin_amdgcn_is_shared((const void*)&S));
#endif
    typedef const double __attribute__((address_space(1))) * GIn;
    typedef double __attribute__((address_space(1))) * GOut;
    const GIn xs_g = (GIn)xs, us_g = (GIn)us, k_g = (GIn)k, K_g = (GIn)K, xg_g = (GIn)P.xg, ug_g = (GIn)P.ug;
    const GOut xs_o = (GOut)xs_new, us_o = (GOut)us_new;
    XlFw& F = S.fw;
    const int n = P.n, m = P.m, T = P.T;
    const int lane = threadIdx.x & 63;
    XlFwWave& V = F.wv[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
    const bool fb = k != nullptr;
    const int ln = lane < n ? lane : n - 1, lm = lane < m ? lane : m - 1; // (lanes beyond n / m repeat the last one)
    const double half = PATH ? 1.0 : 0.5;
    struct Ops { // what step t reads from memory, per lane
        double xo, xg, uo, kk, ug;
        double Kc[XM]; // column ln of K_t
    };
    const double xg0 = PATH ? 0.0 : xg_g[ln], ug0 = PATH ? 0.0 : ug_g[lm];
    auto fetch = [&](Ops& o, int t) {
        const int64_t tc = t < T ? t : T - 1;
        o.uo = us_g[tc * m + lm];
        o.xg = PATH ? xg_g[tc * n + ln] : xg0;
        o.ug = PATH ? ug_g[tc * m + lm] : ug0;
        if (fb) {
            o.xo = xs_g[tc * n + ln];
            o.kk = k_g[tc * m + lm];
#pragma unroll
            for (int a = 0; a < XM; ++a)
                o.Kc[a] = a < m ? K_g[(tc * m + a) * n + ln] : 0.0;
        } else {
            o.xo = 0.0;
            o.kk = 0.0;
        }
    };
    auto put = [&](const Ops& o, double xn, int par) { // the LDS operands of a step whose state is xn
        if (lane < n) {
            V.xn[lane] = xn;
            V.dv[lane] = xn - o.xo;
            V.dg[par][lane] = xn - o.xg;
            if (fb) {
#pragma unroll
                for (int a = 0; a < XM; ++a)
                    if (a < m)
                        V.Kst[a * LDN + lane] = o.Kc[a];
            }
        }
    };
    double xn = ((GIn)x0)[ln];
    const double dt = P.dt, sig_c = P.sig_c;
    // Everything loaded so far is waited for here (vmcnt(0)), before the fetches that stay in flight are issued: a
    // value still pending at the head of the step loop would be waited for in every trip, and a wait for the oldest
    // load is a wait for all of them.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    Ops oa, ob, oc;
    fetch(oa, 0);
    fetch(ob, 1);
    fetch(oc, 2);
    if (lane < n && xs_new)
        xs_o[lane] = xn;
    xl_wave_sync();
    put(oa, xn, 0);
    xl_wave_sync();
    double J = 0.0;
    // `mine`: the operands of step i; `next`: those of step i + 1; `mine` is fetched again for step i + 3
    auto step = [&](int i, int par, Ops& mine, const Ops& next) {
        const double* xv = V.xn;
        const double* dv = V.dv;
        const double* dg = V.dg[par];
        const double* At = F.At + lane;
        const double* Wt = F.Wt + lane;
        const double* Qr = F.Qr + lane;
        const double* Kr = V.Kst + lm * LDN;
        double ax = 0.0, z = 0.0, cq = 0.0, kd = 0.0;
        int j0 = 0;
#pragma unroll 2
        for (; j0 + 8 <= n; j0 += 8) {
            double a_[8], w_[8], q_[8], k_[8], x_[8], d_[8], g_[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                a_[q] = At[(j0 + q) * LDN];
                w_[q] = Wt[(j0 + q) * LDN];
                q_[q] = Qr[(j0 + q) * LDN];
                k_[q] = Kr[j0 + q];
                x_[q] = xv[j0 + q];
                d_[q] = dv[j0 + q];
                g_[q] = dg[j0 + q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool first = q == 0 && j0 == 0; // (a chain starts with a product, not with fma(., ., 0))
                ax = first ? a_[q] * x_[q] : fma(a_[q], x_[q], ax);
                z = first ? w_[q] * x_[q] : fma(w_[q], x_[q], z);
                cq = first ? g_[q] * q_[q] : fma(g_[q], q_[q], cq);
                kd = first ? k_[q] * d_[q] : fma(k_[q], d_[q], kd);
            }
        }
        for (; j0 < n; ++j0) {
            const double a = At[j0 * LDN], w = Wt[j0 * LDN], qq = Qr[j0 * LDN], kk = Kr[j0];
            const double x = xv[j0], d = dv[j0], g = dg[j0];
            ax = j0 == 0 ? a * x : fma(a, x, ax);
            z = j0 == 0 ? w * x : fma(w, x, z);
            cq = j0 == 0 ? g * qq : fma(g, qq, cq);
            kd = j0 == 0 ? kk * d : fma(kk, d, kd);
        }
        // u_new = u + alpha k + K (x_new - x)   (:104, never clamped), or the given control
        const double un = fb ? (mine.uo + alpha * mine.kk) + kd : mine.uo;
        if (lane < m) {
            V.un[lane] = un;
            V.ud[lane] = un - mine.ug;
            if (us_new)
                us_o[(int64_t)i * m + lane] = un;
        }
        if (lane < n)
            V.acc[lane] = cq;
        xl_wave_sync();
        // B u; x_diff^T Q x_diff as (x_diff^T Q) . x_diff; column lm of u_diff^T R
        const double* Bt = F.Bt + lane;
        const double* Rr = F.Rr + lm;
        double bu = 0.0, cu = 0.0;
        {
            int j = 0;
            for (; j + 8 <= m; j += 8) {
                double b_[8], u_[8], d_[8], r_[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    b_[q] = Bt[(j + q) * LDN];
                    u_[q] = V.un[j + q];
                    d_[q] = V.ud[j + q];
                    r_[q] = Rr[(j + q) * LDM];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool first = q == 0 && j == 0;
                    bu = first ? b_[q] * u_[q] : fma(b_[q], u_[q], bu);
                    cu = first ? d_[q] * r_[q] : fma(d_[q], r_[q], cu);
                }
            }
            for (; j < m; ++j) {
                bu = j == 0 ? Bt[j * LDN] * V.un[j] : fma(Bt[j * LDN], V.un[j], bu);
                cu = j == 0 ? V.ud[j] * Rr[j * LDM] : fma(V.ud[j], Rr[j * LDM], cu);
            }
        }
        double sx = V.acc[0] * dg[0];
        j0 = 1;
        if (n >= 8) {
#pragma unroll
            for (int q = 1; q < 8; ++q)
                sx = fma(V.acc[q], dg[q], sx);
            for (j0 = 8; j0 + 8 <= n; j0 += 8) {
                double c_[8], g_[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    c_[q] = V.acc[j0 + q];
                    g_[q] = dg[j0 + q];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    sx = fma(c_[q], g_[q], sx);
            }
        }
        for (; j0 < n; ++j0)
            sx = fma(V.acc[j0], dg[j0], sx);
        // x+ = x + dt ((A x + B u) + c phi(W x))
        const double s = 1.0 + z * z;
        const double phi = z / sqrt(s);
        xn = xn + dt * ((ax + bu) + sig_c * phi);
        if (lane < m)
            V.acu[lane] = cu;
        if (lane < n && xs_new)
            xs_o[(int64_t)(i + 1) * n + lane] = xn;
        put(next, xn, par ^ 1);
        xl_wave_sync();
        double su = 0.0;
        {
            int j = 0;
            for (; j + 8 <= m; j += 8) {
                double c_[8], d_[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    c_[q] = V.acu[j + q];
                    d_[q] = V.ud[j + q];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    su = (q == 0 && j == 0) ? c_[q] * d_[q] : fma(c_[q], d_[q], su);
            }
            for (; j < m; ++j)
                su = j == 0 ? V.acu[j] * V.ud[j] : fma(V.acu[j], V.ud[j], su);
        }
        // l(x, u) (ilqr_cost.py:532-553; PathQRCost :682-701 without the 1/2)
        if (!PATH) {
            sx = sx * half;
            su = su * half;
        }
        J = J + (sx + su);
        asm volatile("" : "+v"(J)); // (the same on every lane, but it lives in a vector register)
        fetch(mine, i + 3);
    };
    for (int i0 = 0; i0 < T; i0 += 6) { // (six = two rounds of the three operand sets: the parity stays a constant)
        if (i0 + 0 < T) step(i0 + 0, 0, oa, ob);
        if (i0 + 1 < T) step(i0 + 1, 1, ob, oc);
        if (i0 + 2 < T) step(i0 + 2, 0, oc, oa);
        if (i0 + 3 < T) step(i0 + 3, 1, oa, ob);
        if (i0 + 4 < T) step(i0 + 4, 0, ob, oc);
        if (i0 + 5 < T) step(i0 + 5, 1, oc, oa);
    }
    xl_wave_sync();
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
            cur = __builtin_amdgcn_readfirstlane(win < cur ? win : win + 1); // (a scalar: it selects the buffers)
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
        const int64_t b = __builtin_amdgcn_readfirstlane(S.next); // (a scalar: every address below is the same on all lanes)
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
    q.n_exhaust = n;
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

#ifdef ILQR_XL_PROF
int ilqr_xl_debug_prof(unsigned long long* host16, int reset)
{
    if (reset) {
        unsigned long long z[16] = {0};
        return xrc(hipMemcpyToSymbol(HIP_SYMBOL(g_xl_prof), z, sizeof(z)));
    }
    return xrc(hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_xl_prof), 16 * sizeof(unsigned long long)));
}
#endif

} // extern "C"
