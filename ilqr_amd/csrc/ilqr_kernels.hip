// HIP kernels (gfx950) and the C ABI of the batched control-limited iLQR solve.
// Mapping: one problem per lane, 64 problems per wavefront, every per-problem
// matrix in registers, trajectories and gains streamed from HBM in batch-minor
// layout ([row, ld] with the problem index innermost) so that each wave access
// is one contiguous 512-byte segment.
#include <hip/hip_runtime.h>

#include <atomic>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <mutex>
#include <type_traits>

#include "../../include/ilqr_hip.h"
#include "ilqr_device.hpp"
#include "ilqr_qp_dense.hpp"

using namespace ilqr;

namespace {

constexpr int WAVE = 64;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int hip_status(hipError_t e) { return e == hipSuccess ? ILQR_OK : ILQR_ERR_HIP; }

struct BoundsP {
    double lo, hi;
};

// ---------------------------------------------------------------- helpers

__device__ __forceinline__ void load5(const double* base, int64_t ld, int64_t row0, int64_t b,
                                      double (&v)[NX])
{
#pragma unroll
    for (int i = 0; i < NX; ++i)
        v[i] = base[(row0 + i) * ld + b];
}

__device__ __forceinline__ void store5(double* base, int64_t ld, int64_t row0, int64_t b,
                                       const double (&v)[NX])
{
#pragma unroll
    for (int i = 0; i < NX; ++i)
        base[(row0 + i) * ld + b] = v[i];
}

// ------------------------------------------------------ layout conversion

// Element (row r, problem b) of a batch-minor array.  ld > 0: plain [R, ld].
// ld == 0: wave-tiled [ceil(B/64), R, 64] -- the 64 problems of one wavefront keep
// all their rows in one contiguous block, so a wave walks its own few-KB region
// sequentially instead of striding across the whole batch (used by the solve
// workspace; DRAM pages and TLB entries are then reused from step to step).
__device__ __forceinline__ int64_t bm_index(int64_t r, int64_t b, int64_t R, int64_t ld)
{
    return ld > 0 ? r * ld + b : ((b >> 6) * R + r) * 64 + (b & 63);
}

// ref [B, R] -> bm: 64x64 tile through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void k_pack(const double* __restrict__ src, double* __restrict__ dst,
                                              int64_t B, int64_t R, int64_t ld)
{
    __shared__ double tile[64][65];
    int64_t b0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)blockIdx.y * 64;
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {
        int64_t b = b0 + j, r = r0 + tx;
        tile[j][tx] = (b < B && r < R) ? src[b * R + r] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        int64_t r = r0 + j, b = b0 + tx;
        if (r < R && b < B)
            dst[bm_index(r, b, R, ld)] = tile[tx][j];
    }
}

// bm -> ref [B, R]; with sel != nullptr problem b reads srcB when sel[b] != 0.
__global__ __launch_bounds__(256) void k_unpack(const double* __restrict__ srcA,
                                                const double* __restrict__ srcB,
                                                const int32_t* __restrict__ sel,
                                                double* __restrict__ dst, int64_t B, int64_t R,
                                                int64_t ld)
{
    __shared__ double tile[64][65];
    int64_t b0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)blockIdx.y * 64;
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int64_t bl = b0 + tx;
    const double* src = srcA;
    if (sel != nullptr && bl < B && sel[bl] != 0)
        src = srcB;
    for (int j = ty; j < 64; j += 4) {
        int64_t r = r0 + j;
        tile[j][tx] = (r < R && bl < B) ? src[bm_index(r, bl, R, ld)] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        int64_t b = b0 + j, r = r0 + tx;
        if (b < B && r < R)
            dst[b * R + r] = tile[tx][j];
    }
}

// The four conversions on either side of a solve in one launch each (eight small launches cost a short
// solve -- a closed-loop re-solve of 15 steps -- a sixth of its time).  Rows [y0[i], y0[i+1]) of the grid's
// y dimension belong to job i.
constexpr int PACK_JOBS = 6;
struct PackJobs {
    const double* src[PACK_JOBS];
    double* dst[PACK_JOBS];
    int64_t R[PACK_JOBS];  // rows of the source, [B, R] ...
    int64_t S[PACK_JOBS];  // ... whose rows are S doubles apart (S >= R)
    int64_t Rd[PACK_JOBS]; // rows per tile of the wave-tiled destination (>= R: the rows land at the top of the tile)
    unsigned y0[PACK_JOBS + 1];
    int n;              // jobs in use
    const int64_t* idx; // not null: problem b of the launch is row idx[b] of the source arrays
};
__global__ __launch_bounds__(256) void k_pack_jobs(PackJobs Jb, int64_t B)
{
    __shared__ double tile[64][65];
    int job = 0;
#pragma unroll
    for (int i = 1; i < PACK_JOBS; ++i)
        job += (i < Jb.n && blockIdx.y >= Jb.y0[i]) ? 1 : 0;
    const double* __restrict__ src = Jb.src[0];
    double* __restrict__ dst = Jb.dst[0];
    int64_t R = Jb.R[0], Rd = Jb.Rd[0], S = Jb.S[0];
    unsigned y0 = Jb.y0[0];
#pragma unroll
    for (int i = 1; i < PACK_JOBS; ++i)
        if (job == i) {
            src = Jb.src[i];
            dst = Jb.dst[i];
            R = Jb.R[i];
            S = Jb.S[i];
            Rd = Jb.Rd[i];
            y0 = Jb.y0[i];
        }
    int64_t b0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)(blockIdx.y - y0) * 64;
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {
        int64_t b = b0 + j, r = r0 + tx;
        double v = 0.0;
        if (b < B && r < R) {
            int64_t row = Jb.idx != nullptr ? Jb.idx[b] : b;
            v = src[row * S + r];
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        int64_t r = r0 + j, b = b0 + tx;
        if (r < R && b < B)
            dst[bm_index(r, b, Rd, 0)] = tile[tx][j];
    }
}

struct UnpackJobs {
    const double* srcA[4];
    const double* srcB[4];
    const int32_t* sel[4];
    double* dst[4];
    int64_t R[4];
    unsigned y0[5];
    const int64_t* idx; // not null: problem b of the launch is row idx[b] of the destination arrays
};
__global__ __launch_bounds__(256) void k_unpack_jobs(UnpackJobs Jb, int64_t B)
{
    __shared__ double tile[64][65];
    int job = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        job += blockIdx.y >= Jb.y0[i] ? 1 : 0;
    const double* srcA = Jb.srcA[0];
    const double* srcB = Jb.srcB[0];
    const int32_t* __restrict__ sel = Jb.sel[0];
    double* __restrict__ dst = Jb.dst[0];
    int64_t R = Jb.R[0];
    unsigned y0 = Jb.y0[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (job == i) {
            srcA = Jb.srcA[i];
            srcB = Jb.srcB[i];
            sel = Jb.sel[i];
            dst = Jb.dst[i];
            R = Jb.R[i];
            y0 = Jb.y0[i];
        }
    int64_t b0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)(blockIdx.y - y0) * 64;
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int64_t bl = b0 + tx;
    const double* src = srcA;
    if (bl < B && sel[bl] != 0)
        src = srcB;
    for (int j = ty; j < 64; j += 4) {
        int64_t r = r0 + j;
        tile[j][tx] = (r < R && bl < B) ? src[bm_index(r, bl, R, 0)] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        int64_t b = b0 + j, r = r0 + tx;
        if (b < B && r < R) {
            int64_t row = Jb.idx != nullptr ? Jb.idx[b] : b;
            dst[row * R + r] = tile[tx][j];
        }
    }
}

// ------------------------------------------------------- rollout and cost

// _dynamics_rollout + _trajectory_cost (ilqr_core.py:69-83, :230-239)
template <class Cost>
__global__ __launch_bounds__(WAVE) void k_rollout_cost(CartpoleP mp, QRCostP Cp, int T, int64_t B,
                                                       int64_t ld, const double* __restrict__ x0,
                                                       const double* __restrict__ us,
                                                       double* __restrict__ xs, double* __restrict__ Jout)
{
    int64_t b = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    if (b >= B)
        return;
    CartpoleC P = cartpole_consts(mp);
    Cost C(Cp);
    double x[NX], xn[NX];
    load5(x0, ld, 0, b, x);
    double J = 0.0;
    for (int t = 0; t < T; ++t) {
        double u = us[(int64_t)t * ld + b];
        store5(xs, ld, (int64_t)t * NX, b, x);
        C.at(t);
        J = J + C.l(x, u);
        cartpole_f(P, x, u, xn);
#pragma unroll
        for (int i = 0; i < NX; ++i)
            x[i] = xn[i];
    }
    store5(xs, ld, (int64_t)T * NX, b, x);
    C.at_terminal(T);
    Jout[b] = J + C.l_terminal(x);
}

// ------------------------------------------------- derivative tensor stream

// One thread per (problem, step): f_x, f_u, l_x, l_u, l_xx, l_uu, l_ux as dense
// batch-minor streams (what _compute_Q_terms asks the model for, ilqr_core.py:189-198).
// Two neighbouring problems per lane, so that every store instruction of a wavefront writes 64 x 16 bytes =
// 1 KB of consecutive memory (the rows of the batch-minor streams are `ld` doubles long and ld is a multiple of
// 64, so an even problem index is 16-byte aligned in every row).
struct DerivOne { // the derivative tensors of one (problem, step), row-major
    double fx[NX * NX], fu[NX], lx[NX], lu, luu;
};
template <class Cost>
__device__ __forceinline__ void deriv_one(const CartpoleC& P, const Cost& C, const double (&x)[NX], double u, int t,
                                          DerivOne& d)
{
    double th = det_atan2(x[2], x[3]);
    double sn, cn;
    det_sincos(th + x[4] * P.dt, sn, cn);
    CartpoleJac J;
    cartpole_jac(P, x, u, sn, cn, J);
#pragma unroll
    for (int i = 0; i < NX * NX; ++i)
        d.fx[i] = 0.0;
    d.fx[0] = 1.0;
    d.fx[1] = P.dt;
    d.fx[1 * NX + 1] = 1.0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            d.fx[(r + 1) * NX + 2 + c] = J.a[r][c];
    d.fu[0] = 0.0;
    d.fu[1] = J.b1;
    d.fu[2] = 0.0;
    d.fu[3] = 0.0;
    d.fu[4] = J.b4;
    C.at(t);
    C.lx(x, d.lx);
    d.lu = C.lu(u);
    d.luu = C.luu();
}
__device__ __forceinline__ void store_pair(double* row, int64_t b, bool two, double v0, double v1)
{
    if (two)
        *reinterpret_cast<double2*>(row + b) = make_double2(v0, v1);
    else
        row[b] = v0;
}
template <class Cost>
__global__ __launch_bounds__(256) void k_deriv(CartpoleP mp, QRCostP Cp, int T, int64_t B, int64_t ld,
                                               const double* __restrict__ xs,
                                               const double* __restrict__ us, ilqr_deriv_bm_t o)
{
    const int64_t b = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x); // problems b and b + 1
    const int t = blockIdx.y;
    if (b >= B)
        return;
    const bool two = b + 1 < B;
    const int64_t b1 = two ? b + 1 : b;
    CartpoleC P = cartpole_consts(mp);
    Cost C(Cp);
    double x0[NX], x1[NX];
    load5(xs, ld, (int64_t)t * NX, b, x0);
    load5(xs, ld, (int64_t)t * NX, b1, x1);
    if (t == T) { // terminal l_x, l_xx (ilqr_core.py:120-121)
        double l0[NX], l1[NX];
        C.at_terminal(T);
        C.lx_terminal(x0, l0);
        C.lx_terminal(x1, l1);
#pragma unroll
        for (int i = 0; i < NX; ++i)
            store_pair(o.lxT + (int64_t)i * ld, b, two, l0[i], l1[i]);
        for (int i = 0; i < NX * NX; ++i)
            store_pair(o.lxxT + (int64_t)i * ld, b, two, Cp.QQt[i], Cp.QQt[i]);
        return;
    }
    DerivOne d0, d1;
    deriv_one(P, C, x0, us[(int64_t)t * ld + b], t, d0);
    deriv_one(P, C, x1, us[(int64_t)t * ld + b1], t, d1);
    const int64_t base = (int64_t)t * NX * NX;
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) {
        store_pair(o.fx + (base + i) * ld, b, two, d0.fx[i], d1.fx[i]);
        store_pair(o.lxx + (base + i) * ld, b, two, Cp.QQ[i], Cp.QQ[i]);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        store_pair(o.fu + ((int64_t)t * NX + i) * ld, b, two, d0.fu[i], d1.fu[i]);
        store_pair(o.lx + ((int64_t)t * NX + i) * ld, b, two, d0.lx[i], d1.lx[i]);
        store_pair(o.lux + ((int64_t)t * NX + i) * ld, b, two, 0.0, 0.0);
    }
    store_pair(o.lu + (int64_t)t * ld, b, two, d0.lu, d1.lu);
    store_pair(o.luu + (int64_t)t * ld, b, two, d0.luu, d1.luu);
}

// ------------------------------------------------------------ backward pass

// One backward sweep for problem b.  Writes k [T, ld], K [T*5, ld]; rows the
// sweep does not reach are zero (np.zeros_like at ilqr_core.py:118-119).
// TRUST_NEXT: take sin/cos of next_theta from xs[t+1] (valid when xs is a rollout
// of us through cartpole_f) instead of re-evaluating them.
// RECORD: also write the per-step boxQP result codes and free masks.
// (Both are compile-time switches so that every loop iteration issues the same
// memory operations: the wait-count bookkeeping then lets the loads of step i-1
// stay in flight while step i computes instead of draining the queue.)
template <bool TRUST_NEXT, bool RECORD, class Cost>
__device__ __forceinline__ int backward_sweep(const CartpoleC& P, const Cost& C, const QPOptsP& QO,
                                              const double* qp_tab, const BoundsP& bd, int T, int64_t ld, int64_t b,
                                              const double* __restrict__ xs,
                                              const double* __restrict__ us, double mu,
                                              double* __restrict__ k, double* __restrict__ K,
                                              BwAcc& S, int& stopped_at, int32_t* qp_result_bm,
                                              int32_t* free_bm)
{
    C.column(b); // (the cost's goal may belong to the problem)
    double xnext[NX];
    load5(xs, ld, (int64_t)T * NX, b, xnext);
    C.at_terminal(T);
    bw_init(C, xnext, S);
    stopped_at = -1;
    double x[NX], u;
    load5(xs, ld, (int64_t)(T - 1) * NX, b, x);
    u = us[(int64_t)(T - 1) * ld + b];
    int rc = BW_OK;
    // Every lane walks all T steps and stores every step: a lane whose sweep has
    // ended (non-PD or non-finite Q_uu) stores the zeros the reference leaves in
    // the steps it never reaches (np.zeros_like, ilqr_core.py:118-119).  Keeping the
    // trip count and the memory operations the same for every lane lets the loads
    // of step i-1 stay in flight behind the stores of step i.
    // results of the previous step, stored one step late (see keep_alive)
    double pend_k = 0.0, pend_K[NX] = {0.0, 0.0, 0.0, 0.0, 0.0};
    int pend_q = -100, pend_f = 0;
    for (int i = T - 1; i >= 0; --i) {
        // prefetch the operands of step i-1 while step i computes (step 0 re-reads itself)
        double xp[NX], up;
        int ip = i > 0 ? i - 1 : 0;
        load5(xs, ld, (int64_t)ip * NX, b, xp);
        up = us[(int64_t)ip * ld + b];
        {   // store step i+1 (at i = T-1 nothing is pending: zeros go to slot T-1 and are
            // overwritten, in program order, by the real step T-1 one iteration later)
            int is = i + 1 < T ? i + 1 : T - 1;
            k[(int64_t)is * ld + b] = pend_k;
            store5(K, ld, (int64_t)is * NX, b, pend_K);
            if (RECORD) {
                qp_result_bm[(int64_t)is * ld + b] = pend_q;
                free_bm[(int64_t)is * ld + b] = pend_f;
            }
        }
        double kk = 0.0, Kr[NX] = {0.0, 0.0, 0.0, 0.0, 0.0};
        int qres = -100;
        bool qfree = false;
        if (rc == BW_OK) {
            double sn, cn;
            if (TRUST_NEXT) {
                sn = xnext[2];
                cn = xnext[3];
            } else {
                double th = det_atan2(x[2], x[3]);
                det_sincos(th + x[4] * P.dt, sn, cn);
            }
            CartpoleJac J;
#ifdef ILQR_ABLATE_JAC // timing diagnostic only (wrong numerics): a Jacobian that costs nothing
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 3; ++c)
                    J.a[r][c] = x[r + 1] + sn;
            J.b1 = x[0];
            J.b4 = cn;
#else
            cartpole_jac(P, x, u, sn, cn, J);
#endif
            C.at(i);
            double kt, Kt[NX];
            int qr;
            bool qf;
            int r = bw_step(C, QO, qp_tab, J, P.dt, x, u, mu, bd.lo, bd.hi, S, kt, Kt, qr, qf);
            if (r == BW_OK) {
                kk = kt;
#pragma unroll
                for (int a = 0; a < NX; ++a)
                    Kr[a] = Kt[a];
                qres = qr;
                qfree = qf;
            } else {
                rc = r;
                if (r == BW_NOT_PD)
                    stopped_at = i;
            }
        }
        keep_alive(pend_k);
#pragma unroll
        for (int a = 0; a < NX; ++a)
            keep_alive(pend_K[a]);
        pend_k = kk;
        pend_q = qres;
        pend_f = qfree ? 1 : 0;
#pragma unroll
        for (int a = 0; a < NX; ++a) {
            pend_K[a] = Kr[a];
            xnext[a] = x[a];
            x[a] = xp[a];
        }
        u = up;
    }
    k[b] = pend_k;
    store5(K, ld, 0, b, pend_K);
    if (RECORD) {
        qp_result_bm[b] = pend_q;
        free_bm[b] = pend_f;
    }
    return rc;
}

template <class Cost, bool RECORD>
__global__ __launch_bounds__(WAVE) void k_backward(CartpoleP mp, QRCostP Cp, QPOptsP QOp, BoundsP bd,
                                                   int T, int64_t B, int64_t ld,
                                                   const double* __restrict__ xs,
                                                   const double* __restrict__ us,
                                                   const double* __restrict__ mu,
                                                   double* __restrict__ k, double* __restrict__ K,
                                                   double* __restrict__ dJ, int32_t* __restrict__ info,
                                                   int32_t* qp_result_bm, int32_t* free_bm)
{
    __shared__ double s_qp_tab[QP_TAB];
    qp_fill_step_table(s_qp_tab, QOp.step_dec);
    int64_t b = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    if (b >= B)
        return;
    CartpoleC P = cartpole_consts(mp);
    Cost C(Cp);
    QPOptsP QO = qp_opts_in_vgprs(QOp);
    BwAcc S;
    int stopped_at;
    int rc = backward_sweep<false, RECORD>(P, C, QO, s_qp_tab, bd, T, ld, b, xs, us, mu[b], k, K, S, stopped_at,
                                   qp_result_bm, free_bm);
    dJ[b] = S.dJ1;
    dJ[ld + b] = S.dJ2;
    info[b] = stopped_at;
    info[ld + b] = S.qp_iter;
    info[2 * ld + b] = S.qp_backtrack;
    info[3 * ld + b] = S.n_clamped;
    info[4 * ld + b] = rc;
}

// The staged variant: the sweep reads the derivative tensors a derivative kernel has written
// (f_x, f_u, l_x, l_u, l_xx, l_uu, l_ux per step and the terminal l_x, l_xx) instead of
// evaluating the model: 67 doubles = 536 bytes per (problem, step) streamed from HBM.
template <bool RECORD>
__global__ __launch_bounds__(WAVE) void k_backward_staged(QPOptsP QOp, BoundsP bd, int T, int64_t B, int64_t ld,
                                                          const double* __restrict__ us,
                                                          const double* __restrict__ mu, ilqr_deriv_bm_t d,
                                                          double* __restrict__ k, double* __restrict__ K,
                                                          double* __restrict__ dJ, int32_t* __restrict__ info,
                                                          int32_t* qp_result_bm, int32_t* free_bm)
{
    __shared__ double s_qp_tab[QP_TAB];
    qp_fill_step_table(s_qp_tab, QOp.step_dec);
    int64_t b = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    if (b >= B)
        return;
    QPOptsP QO = qp_opts_in_vgprs(QOp);
    BwAcc S;
    // V_x, V_xx = terminal l_x, l_xx (ilqr_core.py:120-121)
    load5(d.lxT, ld, 0, b, S.Vx);
#pragma unroll
    for (int i = 0; i < NX * NX; ++i)
        S.Vxx[i] = d.lxxT[(int64_t)i * ld + b];
    S.dJ1 = 0.0;
    S.dJ2 = 0.0;
    S.qp_iter = S.qp_backtrack = S.n_clamped = 0;
    const double m = mu[b];
    int stopped_at = -1, rc = BW_OK;
    for (int i = T - 1; i >= 0; --i) {
        double kk = 0.0, Kr[NX] = {0.0, 0.0, 0.0, 0.0, 0.0};
        int qres = -100;
        bool qfree = false;
        if (rc == BW_OK) {
            double fx[NX * NX], lxx[NX * NX], fu[NX], lx[NX], lux[NX];
            int64_t base = (int64_t)i * NX * NX;
#pragma unroll
            for (int e = 0; e < NX * NX; ++e) {
                fx[e] = d.fx[(base + e) * ld + b];
                lxx[e] = d.lxx[(base + e) * ld + b];
            }
            load5(d.fu, ld, (int64_t)i * NX, b, fu);
            load5(d.lx, ld, (int64_t)i * NX, b, lx);
            load5(d.lux, ld, (int64_t)i * NX, b, lux);
            double lu = d.lu[(int64_t)i * ld + b], luu = d.luu[(int64_t)i * ld + b];
            double u = us[(int64_t)i * ld + b];
            double kt, Kt[NX];
            int qr;
            bool qf;
            int r = bw_step_dense(QO, s_qp_tab, fx, fu, lx, lu, lxx, luu, lux, u, m, bd.lo, bd.hi, S, kt, Kt, qr, qf);
            if (r == BW_OK) {
                kk = kt;
#pragma unroll
                for (int a = 0; a < NX; ++a)
                    Kr[a] = Kt[a];
                qres = qr;
                qfree = qf;
            } else {
                rc = r;
                if (r == BW_NOT_PD)
                    stopped_at = i;
            }
        }
        k[(int64_t)i * ld + b] = kk;
        store5(K, ld, (int64_t)i * NX, b, Kr);
        if (RECORD) {
            qp_result_bm[(int64_t)i * ld + b] = qres;
            free_bm[(int64_t)i * ld + b] = qfree ? 1 : 0;
        }
    }
    dJ[b] = S.dJ1;
    dJ[ld + b] = S.dJ2;
    info[b] = stopped_at;
    info[ld + b] = S.qp_iter;
    info[2 * ld + b] = S.qp_backtrack;
    info[3 * ld + b] = S.n_clamped;
    info[4 * ld + b] = rc;
}

// ------------------------------------------------------------- forward pass

// _forward_pass + _trajectory_cost for one (problem, alpha)   (ilqr_core.py:85-107, :230-239).
// xs_new / us_new may be null (cost only).
__device__ __forceinline__ void pair_barrier();
// SYNC: the wavefront attends one workgroup hand-over per step (it runs beside a rollout that two other wavefronts
// share, see forward_rollout_master).
template <bool STORE, class Cost, bool SYNC = false>
__device__ __forceinline__ double forward_rollout(const CartpoleC& P, const Cost& C, int T,
                                                  int64_t ld, int64_t b,
                                                  const double* __restrict__ xs,
                                                  const double* __restrict__ us,
                                                  const double* __restrict__ k,
                                                  const double* __restrict__ K, double alpha,
                                                  double* __restrict__ xs_new,
                                                  double* __restrict__ us_new)
{
    C.column(b); // (the cost's goal may belong to the problem)
    // The operands of a step (old x, u, k, K: 12 numbers) are loaded TWO steps ahead: with one
    // wave per SIMD the bytes in flight are what sets the achieved bandwidth of a large batch.
    struct Ops {
        double xo[NX], Kr[NX], u, kk;
    };
    auto fetch = [&](int t, Ops& o) {
        load5(xs, ld, (int64_t)t * NX, b, o.xo);
        load5(K, ld, (int64_t)t * NX, b, o.Kr);
        o.u = us[(int64_t)t * ld + b];
        o.kk = k[(int64_t)t * ld + b];
    };
    double xn[NX], xt[NX];
    load5(xs, ld, 0, b, xn);
    double J = 0.0;
    Ops cur, nxt;
    fetch(0, cur);
    fetch(T > 1 ? 1 : 0, nxt);
    // (x_t, u_t) of the previous step, stored one step late (see keep_alive)
    double pend_x[NX], pend_u = 0.0;
#pragma unroll
    for (int a = 0; a < NX; ++a)
        pend_x[a] = xn[a];
#pragma unroll 2
    for (int t = 0; t < T; ++t) {
        Ops far; // step t+2 (the last steps re-read the last one)
        fetch(t + 2 < T ? t + 2 : T - 1, far);
        if (STORE) { // step t-1 (at t = 0: x_0 and a placeholder u_0, rewritten at t = 1)
            int ts = t > 0 ? t - 1 : 0;
            store5(xs_new, ld, (int64_t)ts * NX, b, pend_x);
            us_new[(int64_t)ts * ld + b] = pend_u;
        }
        double un = fw_control(xn, cur.xo, cur.u, cur.kk, cur.Kr, alpha);
        C.at(t);
        J = J + C.l(xn, un);
        cartpole_f(P, xn, un, xt);
        if (SYNC)
            pair_barrier();
        if (STORE) {
#pragma unroll
            for (int a = 0; a < NX; ++a)
                keep_alive(pend_x[a]);
            keep_alive(pend_u);
        }
        pend_u = un;
#pragma unroll
        for (int a = 0; a < NX; ++a) {
            pend_x[a] = xn[a];
            xn[a] = xt[a];
        }
        cur = nxt;
        nxt = far;
    }
    if (STORE) {
        store5(xs_new, ld, (int64_t)(T - 1) * NX, b, pend_x);
        us_new[(int64_t)(T - 1) * ld + b] = pend_u;
        store5(xs_new, ld, (int64_t)T * NX, b, xn);
    }
    C.at_terminal(T);
    return J + C.l_terminal(xn);
}

struct AlphasP {
    double a[32];
};

// Lanes = problems, blockIdx.y = alpha: every (problem, alpha) pair in parallel.
template <class Cost, bool STORE>
__global__ __launch_bounds__(WAVE) void k_forward_ls(CartpoleP mp, QRCostP Cp, AlphasP al, int T,
                                                     int64_t B, int64_t ld,
                                                     const double* __restrict__ xs,
                                                     const double* __restrict__ us,
                                                     const double* __restrict__ k,
                                                     const double* __restrict__ K,
                                                     double* __restrict__ Jnew,
                                                     double* __restrict__ xs_new,
                                                     double* __restrict__ us_new)
{
    int64_t b = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    int ai = blockIdx.y;
    if (b >= B)
        return;
    CartpoleC P = cartpole_consts(mp);
    Cost C(Cp);
    double* xo = xs_new ? xs_new + (int64_t)ai * (T + 1) * NX * ld : nullptr;
    double* uo = us_new ? us_new + (int64_t)ai * T * ld : nullptr;
    Jnew[(int64_t)ai * ld + b] = forward_rollout<STORE>(P, C, T, ld, b, xs, us, k, K, al.a[ai], xo, uo);
}

// ------------------------------------------- sweeps shared by two or three wavefronts
//
// While a batch leaves SIMDs idle the solve takes as long as ONE wavefront needs for its instruction stream;
// most of that is the backward sweep's serial chain
//     V  ->  Q_uu, Q_u  ->  Cholesky, boxQP  ->  k, K  ->  V of the step before,
// of which the boxQP's three passes alone are ~450 instructions that depend on one another (a dependent fp64
// operation issues every ~8 cycles).  With SIMDs to spare the sweep is cut between two wavefronts that hold
// the same problems lane for lane and hand a few numbers per lane back and forth through LDS:
//   master: from rows 1 and 4 of V (the non-zero rows of f_u) Q_uu and Q_u, the PD test, Cholesky, boxQP,
//           Delta_J (the part of ilqr_core.py:137-156, :175-176 that the rest of the step waits for);
//   helper: Jacobian, Q_x, Q_xx, Q_ux (:206-212), then K and the value update (:157-169).
// Two things take work out of the chain itself.
//  * The master's next step needs six numbers of the value update only (entries (1,1), (1,4), (4,1), (4,4)
//    of V_xx and V_x[1], V_x[4]); the helper forms and hands over those first and completes the update while
//    the master is in its boxQP again.
//  * The boxQP ends on its common path (jump in pass 0, full Newton step in pass 1, a termination test in
//    pass 2) in almost every call, and the outcome of that path follows from ~35 dependent operations
//    (boxqp1_guess).  The master hands that GUESS to the helper at once and only then runs boxqp1 in full,
//    which takes as long as the helper needs for the six numbers.  If the verified outcome differs from the
//    guess in any lane (wave-uniform flag), the helper recomputes its six numbers from the verified outcome
//    in one extra hand-over; results therefore never depend on a guess.
// Hand-overs per step: #2 master -> helper (the guess), #1 helper -> master (six numbers + f_u of the step
// before + the master's verdict on the previous guess), #1b only after a wrong guess.  Every number is
// computed by the expression the one-wavefront sweep uses, hence the same bits.
// With a third wavefront per group (pair_verifier_sweep) the verification leaves the master as well: the verifier
// reads the same hand-over, runs boxqp1, meets the other two at #2 between its first and second pass and raises
// the flag; the master, idle after its guess, forms the Jacobians two steps ahead for the helper (X.ja).
// The first rollout of an iteration is shared too (forward_rollout_master / forward_rollout_helper): the helper
// runs the chain atan2 -> next angle -> sin, cos, the master everything else of a step.
struct PairX {
    double v[5][64];          // helper -> master, for the master's next step: Q_uu, Q_u, its Cholesky factor L, 1 / L, the warm start
    int st[64];               // ... and BW_OK / BW_NOT_PD / BW_LINALG (the PD test of ilqr_core.py:137-140)
    double kk[64], rdiag[64]; // master -> helper, guessed: boxQP solution, reciprocal of the 1x1 factor
    int free_[64], ok[64];    // guessed free mask; ok = 0: the lane's sweep has ended (not PD / non-finite Q_uu), never a guess
    double kk2[64], rdiag2[64]; // master -> helper, verified (read only after a wrong guess)
    int free2[64];
    int fix;                  // the guess of the step just verified was wrong in some lane
    double mu[64];            // per sweep, master -> helper
    int act[64], cur[64], gnew[64];
    int cmd;                  // 1 = run a backward sweep, 2 = run the angle chain of a rollout, 0 = leave
    double dJ[2][64];         // per sweep, verifier -> master (three wavefronts): Delta_J1, Delta_J2 ...
    int cnt[3][64];           // ... and the boxQP counters
    double ja[14][64];        // master -> helper (three wavefronts): the Jacobian of the step after next
    // three wavefronts: further step sizes tried by the verifier beside the first rollout, for up to SPEC_MAX problems
    // that went past the first step size in the iteration before
    int spec_n, spec_col[2], spec_cur[2], spec_gcur[2];
    double specJ[2][32];
    double fw[2][3][64];      // shared rollout (two sets, used in turn): helper -> master sin, cos of the next angle; master -> helper its rate
};

// LDS hand-over between the wavefronts of a group: waits for LDS only, so global loads issued ahead
// of the barrier stay in flight across it.
__device__ __forceinline__ void pair_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#ifdef ILQR_PAIR_PROF // diagnostic build only: shader clocks of the two halves of the shared backward sweep
static __device__ unsigned long long g_pair_prof[4][8];
#define PP_DECL unsigned long long pp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PP_T(var) const long long var = __builtin_amdgcn_s_memtime()
#define PP_ADD(role, slot, a, bq) pp_acc[slot] += (unsigned long long)((bq) - (a))
#define PP_COUNT(role, slot) pp_acc[slot] += 1ull
#define PP_FLUSH(role)                                                                \
    if (lane == 0 && blockIdx.x == 0)                                                  \
        for (int q_ = 0; q_ < 8; ++q_)                                                 \
            atomicAdd(&g_pair_prof[role][q_], pp_acc[q_])
#else
#define PP_DECL
#define PP_T(var)
#define PP_ADD(role, slot, a, bq)
#define PP_COUNT(role, slot)
#define PP_FLUSH(role)
#endif

// The accepted-step rollout (alpha_0, with stores) shared by two wavefronts that hold the same problems lane for
// lane.  Within a step the next angle (atan2 of the stored sin/cos, plus rate x dt, then sin/cos: one long chain
// of dependent operations) and the next velocities (control law, accelerations) do not depend on one another;
// each needs the other's result of the step before.  The master keeps everything forward_rollout does except the
// angle; the helper runs the angle chain; they swap three numbers per lane and step through LDS.
template <class Cost>
__device__ __forceinline__ double forward_rollout_master(const CartpoleC& P, const Cost& C, int T, int64_t ld, int64_t b,
                                                         PairX& X, int lane, const double* __restrict__ xs,
                                                         const double* __restrict__ us, const double* __restrict__ k,
                                                         const double* __restrict__ K, double alpha,
                                                         double* __restrict__ xs_new, double* __restrict__ us_new)
{
    C.column(b); // (the cost's goal may belong to the problem)
    struct Ops {
        double xo[NX], Kr[NX], u, kk;
    };
    auto fetch = [&](int t, Ops& o) {
        load5(xs, ld, (int64_t)t * NX, b, o.xo);
        load5(K, ld, (int64_t)t * NX, b, o.Kr);
        o.u = us[(int64_t)t * ld + b];
        o.kk = k[(int64_t)t * ld + b];
    };
    double xn[NX], xt[NX];
    load5(xs, ld, 0, b, xn);
    double J = 0.0;
    Ops cur, nxt;
    fetch(0, cur);
    fetch(T > 1 ? 1 : 0, nxt);
    double pend_x[NX], pend_u = 0.0;
#pragma unroll
    for (int a = 0; a < NX; ++a)
        pend_x[a] = xn[a];
    PP_DECL;
    PP_T(tf0);
#pragma unroll 2
    for (int t = 0; t < T; ++t) {
        Ops far;
        fetch(t + 2 < T ? t + 2 : T - 1, far);
        {
            int ts = t > 0 ? t - 1 : 0;
            store5(xs_new, ld, (int64_t)ts * NX, b, pend_x);
            us_new[(int64_t)ts * ld + b] = pend_u;
        }
        double un = fw_control(xn, cur.xo, cur.u, cur.kk, cur.Kr, alpha);
        cartpole_f_velocities(P, xn, un, xt[0], xt[1], xt[4]);
        X.fw[t & 1][2][lane] = xt[4];
        C.at(t);
        J = J + C.l(xn, un);
        PP_T(tf1);
        pair_barrier();
        PP_T(tf2);
        PP_ADD(3, 1, tf1, tf2);
        xt[2] = X.fw[t & 1][0][lane];
        xt[3] = X.fw[t & 1][1][lane];
#pragma unroll
        for (int a = 0; a < NX; ++a)
            keep_alive(pend_x[a]);
        keep_alive(pend_u);
        pend_u = un;
#pragma unroll
        for (int a = 0; a < NX; ++a) {
            pend_x[a] = xn[a];
            xn[a] = xt[a];
        }
        cur = nxt;
        nxt = far;
    }
    PP_T(tf3);
    PP_ADD(3, 0, tf0, tf3);
    PP_FLUSH(3);
    store5(xs_new, ld, (int64_t)(T - 1) * NX, b, pend_x);
    us_new[(int64_t)(T - 1) * ld + b] = pend_u;
    store5(xs_new, ld, (int64_t)T * NX, b, xn);
    C.at_terminal(T);
    return J + C.l_terminal(xn);
}

__device__ __forceinline__ void forward_rollout_helper(const CartpoleC& P, int T, int64_t ld, int64_t b, PairX& X, int lane,
                                                       const double* __restrict__ xs)
{
    double s = xs[2 * ld + b], c = xs[3 * ld + b], thd = xs[4 * ld + b];
    PP_DECL;
    PP_T(tf0);
    for (int t = 0; t < T; ++t) {
        double sn, cn;
        cartpole_f_angle(P, s, c, thd, sn, cn);
        X.fw[t & 1][0][lane] = sn;
        X.fw[t & 1][1][lane] = cn;
        PP_T(tf1);
        pair_barrier();
        PP_T(tf2);
        PP_ADD(3, 3, tf1, tf2);
        thd = X.fw[t & 1][2][lane];
        s = sn;
        c = cn;
    }
    PP_T(tf3);
    PP_ADD(3, 2, tf0, tf3);
    PP_FLUSH(3);
}


// Master half of one backward sweep.  act: the lane takes part (its problem still needs this backward pass).
// S receives Delta_J and the boxQP counters; k [T, ld] is written for the lanes that take part (zeros where
// the sweep did not reach).
// VERIFY = false: a third wavefront runs the verification (pair_verifier_sweep); the master only guesses.
template <class Cost, bool VERIFY>
__device__ __forceinline__ int pair_master_sweep(const Cost& C, const QPOptsP& QO, const double* qp_tab,
                                                 const BoundsP& bd, int T, int64_t ld, int64_t b, PairX& X, int lane,
                                                 bool act, const double* __restrict__ us, double mu,
                                                 double* __restrict__ k, BwAcc& S, int& stopped_at,
                                                 const CartpoleC& P, const double* __restrict__ xs)
{
    C.column(b); // (the cost's goal may belong to the problem)
    S.dJ1 = 0.0;
    S.dJ2 = 0.0;
    S.qp_iter = S.qp_backtrack = S.n_clamped = 0;
    stopped_at = -1;
    int rc = BW_OK;
    double u = us[(int64_t)(T - 1) * ld + b];
    PP_DECL;
#ifdef ILQR_PAIR_PROF
    const int O_n_exhaust_dbg = QO.n_exhaust;
#endif
    if (lane == 0)
        X.fix = 0;
    bool fix = false; // wave-uniform: the guess of the step before was wrong in some lane
    // three wavefronts: the master, idle while the other two work on its guess, forms the Jacobian of step i-2
    // for the helper (which reads it after hand-over #1 of step i-1 and has those of steps T-1, T-2 itself)
    double xm[NX];
    if (!VERIFY)
        load5(xs, ld, (int64_t)(T > 1 ? T - 2 : 0) * NX, b, xm);
    for (int i = T - 1; i >= 0; --i) {
        const int ip = i > 0 ? i - 1 : 0;
        const double up = us[(int64_t)ip * ld + b]; // operand of the step before, loaded ahead
        double xj[NX], uj = 0.0;
        if (!VERIFY) {
            const int ij = i > 1 ? i - 2 : 0;
            load5(xs, ld, (int64_t)ij * NX, b, xj);
            uj = us[(int64_t)ij * ld + b];
        }
        PP_T(t0);
        pair_barrier(); // #1: the helper's numbers of step i are in X (formed from the guess of step i+1) ...
        if (!VERIFY)
            fix = X.fix != 0; // the verifier's flag (it also clears it)
        if (fix) { // ... which was wrong: the helper is forming them again from the verified outcome
            pair_barrier(); // #1b
            if (VERIFY && lane == 0)
                X.fix = 0;
            fix = false;
        }
        PP_T(t1);
        PP_ADD(0, 0, t0, t1);
        const double Quu = X.v[0][lane], Qu = X.v[1][lane], L = X.v[2][lane], rL = X.v[3][lane], k0 = X.v[4][lane];
        const int st = X.st[lane];
        double kk = 0.0, rdiag = 1.0;
        int fr = 0, ok = 0;
        bool go = act && rc == BW_OK;
        if (go) {
            if (st != BW_OK) { // the helper's PD test: np.all(np.linalg.eigvals(Q_uu) > 0)   (:137-140)
                rc = st;
                stopped_at = st == BW_NOT_PD ? i : stopped_at;
                go = false;
            } else {
                bool fg;
                boxqp1_guess(QO, qp_tab, Quu, rL, Qu, bd.lo - u, bd.hi - u, k0, kk, fg);
                rdiag = rL;
                fr = fg ? 1 : 0;
                ok = 1;
            }
        }
        X.kk[lane] = kk;
        X.rdiag[lane] = rdiag;
        X.free_[lane] = fr;
        X.ok[lane] = ok;
        PP_T(t2);
        PP_ADD(0, 1, t1, t2);
        pair_barrier(); // #2: the helper goes on with K and the value update, on the guess
        __builtin_amdgcn_sched_barrier(0); // (the verification below must not be scheduled in front of the hand-over)
        PP_T(t3);
        PP_ADD(0, 2, t2, t3);
        bool wrong = false;
        if (VERIFY && go) {
            QP1Out qp;
            boxqp1(QO, qp_tab, Quu, L, rL, Qu, bd.lo - u, bd.hi - u, k0, qp); // :152-156
            // (the factor only enters K, and K is zero for a clamped control)
            wrong = !(qp.x == kk) | ((qp.free_ ? 1 : 0) != fr) | (qp.free_ & !(qp.rdiag == rdiag));
#ifdef ILQR_PAIR_PROF
            {
                const bool exh = qp.n_backtrack >= O_n_exhaust_dbg;
                const int cls = qp.n_iter == 0 ? 0 : (qp.n_iter == 1 ? 1 : (exh ? 2 : 3));
                for (int c = 0; c < 4; ++c) {
                    unsigned long long m = __ballot(wrong && cls == c);
                    if (m != 0ull && lane == 0 && blockIdx.x == 0)
                        atomicAdd(&g_pair_prof[1][4 + c], (unsigned long long)__popcll(m));
                }
            }
#endif
            kk = qp.x;
            rdiag = qp.rdiag;
            fr = qp.free_ ? 1 : 0;
            S.qp_iter += qp.n_iter;
            S.qp_backtrack += qp.n_backtrack;
            S.n_clamped += qp.free_ ? 0 : 1;
            // Delta_J1 += k.Q_u ; Delta_J2 += 0.5 k.(Q_uu k)                 (:175-176)
            S.dJ1 = S.dJ1 + kk * Qu;
            S.dJ2 = S.dJ2 + (kk * (Quu * kk)) * 0.5;
        }
        if (!VERIFY) {
            CartpoleJac Jm;
            cartpole_jac(P, xj, uj, xm[2], xm[3], Jm);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    X.ja[r * 3 + c][lane] = Jm.a[r][c];
            X.ja[12][lane] = Jm.b1;
            X.ja[13][lane] = Jm.b4;
#pragma unroll
            for (int a = 0; a < NX; ++a)
                xm[a] = xj[a];
        }
        if (VERIFY) {
            if (__ballot(wrong) != 0ull) { // the verified outcome of every lane, and the flag that makes the helper use it
                PP_COUNT(0, 4);
                X.kk2[lane] = kk;
                X.rdiag2[lane] = rdiag;
                X.free2[lane] = fr;
                if (lane == 0)
                    X.fix = 1;
                fix = true;
            }
            if (act)
                k[(int64_t)i * ld + b] = kk;
        }
        u = up;
        PP_T(t4);
        PP_ADD(0, 3, t3, t4);
    }
    pair_barrier(); // the verdict on the guess of step 0 is in X
    if (!VERIFY) { // ... and the verifier's sums
        S.dJ1 = X.dJ[0][lane];
        S.dJ2 = X.dJ[1][lane];
        S.qp_iter = X.cnt[0][lane];
        S.qp_backtrack = X.cnt[1][lane];
        S.n_clamped = X.cnt[2][lane];
    }
    PP_FLUSH(0);
    return rc;
}

// Third wavefront of a group: runs boxqp1 in full on the numbers the helper handed over, compares the outcome
// with the master's guess, raises the flag that makes the helper redo a step after a wrong guess, stores k and
// accumulates Delta_J and the boxQP counters (handed to the master at the end of the sweep).  It meets the
// other two at every hand-over; the hand-over of the guess (#2) falls between the boxQP's first and second pass.
__device__ __forceinline__ void pair_verifier_sweep(const QPOptsP& QO, const double* qp_tab, const BoundsP& bd, int T,
                                                    int64_t ld, int64_t b, PairX& X, int lane, bool act,
                                                    const double* __restrict__ us, double* __restrict__ k)
{
    double dJ1 = 0.0, dJ2 = 0.0;
    int qp_iter = 0, qp_backtrack = 0, n_clamped = 0;
    bool ended = false; // the lane's sweep has ended (not PD / non-finite Q_uu)
    double u = us[(int64_t)(T - 1) * ld + b];
    bool fix = false;
    PP_DECL;
    for (int i = T - 1; i >= 0; --i) {
        const int ip = i > 0 ? i - 1 : 0;
        const double up = us[(int64_t)ip * ld + b];
        PP_T(t0);
        pair_barrier(); // #1
        if (fix) {
            pair_barrier(); // #1b
            if (lane == 0)
                X.fix = 0;
            fix = false;
        }
        PP_T(t1);
        PP_ADD(2, 0, t0, t1);
        const double Quu = X.v[0][lane], Qu = X.v[1][lane], L = X.v[2][lane], rL = X.v[3][lane], k0 = X.v[4][lane];
        const int st = X.st[lane];
        ended = ended || st != BW_OK;
        const bool go = act && !ended;
        QP1Out qp;
        long long tmid0 = 0, tmid1 = 0;
        (void)tmid0;
        (void)tmid1;
        // every lane runs the boxQP (the hand-over inside it is the whole wavefront's); lanes without a QP get one
        // that ends at once, so that no leftover number of theirs sends the wave into a slow path
        boxqp1(QO, qp_tab, go ? Quu : 1.0, go ? L : 1.0, go ? rL : 1.0, go ? Qu : 0.0, go ? bd.lo - u : -1.0,
               go ? bd.hi - u : 1.0, go ? k0 : 0.0, qp, [&]() {
            __builtin_amdgcn_sched_barrier(0);
#ifdef ILQR_PAIR_PROF
            tmid0 = __builtin_amdgcn_s_memtime();
#endif
            pair_barrier(); // #2: the master's guess is in X
#ifdef ILQR_PAIR_PROF
            tmid1 = __builtin_amdgcn_s_memtime();
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef ILQR_PAIR_PROF
        pp_acc[1] += (unsigned long long)(tmid0 - t1);
        pp_acc[2] += (unsigned long long)(tmid1 - tmid0);
#endif
        double kk = 0.0, rdiag = 1.0;
        int fr = 0;
        bool wrong = false;
        if (go) {
            const double kg = X.kk[lane], rg = X.rdiag[lane];
            const int fg = X.free_[lane];
            // (the factor only enters K, and K is zero for a clamped control)
            wrong = !(qp.x == kg) | ((qp.free_ ? 1 : 0) != fg) | (qp.free_ & !(qp.rdiag == rg));
            kk = qp.x;
            rdiag = qp.rdiag;
            fr = qp.free_ ? 1 : 0;
            qp_iter += qp.n_iter;
            qp_backtrack += qp.n_backtrack;
            n_clamped += qp.free_ ? 0 : 1;
            dJ1 = dJ1 + kk * Qu;
            dJ2 = dJ2 + (kk * (Quu * kk)) * 0.5;
        }
        if (__ballot(wrong) != 0ull) {
            PP_COUNT(2, 4);
            X.kk2[lane] = kk;
            X.rdiag2[lane] = rdiag;
            X.free2[lane] = fr;
            if (lane == 0)
                X.fix = 1;
            fix = true;
        }
        if (act)
            k[(int64_t)i * ld + b] = kk;
        u = up;
#ifdef ILQR_PAIR_PROF
        pp_acc[3] += (unsigned long long)(__builtin_amdgcn_s_memtime() - tmid1);
#endif
    }
    X.dJ[0][lane] = dJ1;
    X.dJ[1][lane] = dJ2;
    X.cnt[0][lane] = qp_iter;
    X.cnt[1][lane] = qp_backtrack;
    X.cnt[2][lane] = n_clamped;
    pair_barrier(); // the verdict on the guess of step 0, and the sums, are in X
    PP_FLUSH(2);
}

// Helper half of the same sweep.  K [T*5, ld] is written for the lanes that take part.
// JAC_FROM_MASTER (three wavefronts): the Jacobians of steps T-3 ... 0 come from the master through X.ja.
template <class Cost, bool JAC_FROM_MASTER>
__device__ __forceinline__ void pair_helper_sweep(const CartpoleC& P, const Cost& C, int T, int64_t ld, int64_t b,
                                                  PairX& X, int lane, bool act, const double* __restrict__ xs,
                                                  const double* __restrict__ us, double mu, double* __restrict__ K)
{
    C.column(b); // (the cost's goal may belong to the problem)
    BwAcc S;
    double xnext[NX];
    load5(xs, ld, (int64_t)T * NX, b, xnext);
    C.at_terminal(T);
    bw_init(C, xnext, S);
    double x[NX], u, xq[NX], uq; // operands of step i and of step i-1
    load5(xs, ld, (int64_t)(T - 1) * NX, b, x);
    u = us[(int64_t)(T - 1) * ld + b];
    {
        const int iq = T > 1 ? T - 2 : 0;
        load5(xs, ld, (int64_t)iq * NX, b, xq);
        uq = us[(int64_t)iq * ld + b];
    }
    CartpoleJac J; // of step i: sin/cos of the next angle are entries 2, 3 of the next state
    cartpole_jac(P, x, u, xnext[2], xnext[3], J);
    CartpoleJac Jn; // of step i-1
    if (JAC_FROM_MASTER)
        cartpole_jac(P, xq, uq, x[2], x[3], Jn);
    const double dt = P.dt;
    bool live = act, live_before = act; // live_before: before the outcome of the step in hand was applied
    PP_DECL;
    // Loop-carried across the hand-over: the Q-terms, gains and boxQP outcome of the step before, whose value
    // update (all but the six numbers the master needed) and K store are completed after barrier #1.
    bool pending = false;
    double Qx[NX], Qxx[NX * NX], Qux[NX], Kr[NX], T1[NX], kk = 0.0, Qu = 0.0, Quu = 1.0;
    // K, T1 = K Q_uu from the boxQP outcome (guessed or verified)   (:157-160)
    auto gains = [&](double rdiag, bool fr) {
#pragma unroll
        for (int bb = 0; bb < NX; ++bb) {
            Kr[bb] = (live && fr) ? ((Qux[bb] * rdiag) * rdiag) * -1 : 0.0;
            T1[bb] = Kr[bb] * Quu;
        }
    };
    // What the master's next step waits for: from the six numbers of V that f_u (non-zero in rows 1 and 4) touches,
    // Q_u = l_u + f_u^T V_x ; reg = V_xx + mu I ; M2 = f_u^T reg ; Q_uu = l_uu + M2 f_u   (:207, :210-212), the PD
    // test, the Cholesky factor and the (positive-sign) boxQP warm start (:137-145).  lu_n, luu_n, J: of that step.
    auto publish = [&](double v11, double v14, double v41, double v44, double vx1, double vx4, double lu_n,
                       double luu_n, double b1, double b4) {
        const double Qu_n = lu_n + fma(b4, vx4, b1 * vx1);
        const double M2_1 = fma(b4, v41 + 0.0, b1 * (v11 + mu));
        const double M2_4 = fma(b4, v44 + mu, b1 * (v14 + 0.0));
        const double Quu_n = luu_n + fma(M2_4, b4, M2_1 * b1);
        const int st = !(fabs(Quu_n) < __builtin_inf()) ? BW_LINALG : (!(Quu_n > 0.0) ? BW_NOT_PD : BW_OK);
        const double L = sqrt(Quu_n);
        const double rL = 1.0 / L;
        X.v[0][lane] = Quu_n;
        X.v[1][lane] = Qu_n;
        X.v[2][lane] = L;
        X.v[3][lane] = rL;
        X.v[4][lane] = (Qu_n * rL) * rL;
        X.st[lane] = st;
    };
    // ... after step i (whose Q-terms, gains and k are in hand), for step i-1 (operands xq, uq; its Jacobian is in J)
    auto hand_over = [&](int i, double b1, double b4) {
        const double vn11 = ((Qxx[1 * NX + 1] + T1[1] * Kr[1]) + Kr[1] * Qux[1]) + Qux[1] * Kr[1];
        const double vn14 = ((Qxx[1 * NX + 4] + T1[1] * Kr[4]) + Kr[1] * Qux[4]) + Qux[1] * Kr[4];
        const double vn41 = ((Qxx[4 * NX + 1] + T1[4] * Kr[1]) + Kr[4] * Qux[1]) + Qux[4] * Kr[1];
        const double vn44 = ((Qxx[4 * NX + 4] + T1[4] * Kr[4]) + Kr[4] * Qux[4]) + Qux[4] * Kr[4];
        const double v14 = 0.5 * (vn14 + vn41);
        const double vx1 = ((Qx[1] + T1[1] * kk) + Kr[1] * Qu) + Qux[1] * kk;
        const double vx4 = ((Qx[4] + T1[4] * kk) + Kr[4] * Qu) + Qux[4] * kk;
        C.at(i > 0 ? i - 1 : 0);
        publish(0.5 * (vn11 + vn11), v14, v14, 0.5 * (vn44 + vn44), vx1, vx4, C.lu(uq), C.luu(), b1, b4);
    };
    C.at(T - 1);
    publish(S.Vxx[1 * NX + 1], S.Vxx[1 * NX + 4], S.Vxx[4 * NX + 1], S.Vxx[4 * NX + 4], S.Vx[1], S.Vx[4], C.lu(u), C.luu(),
            J.b1, J.b4);
    for (int i = T - 1; i >= 0; --i) {
        PP_T(t0);
        pair_barrier(); // #1: the master has verified its guess of step i+1
        if (X.fix != 0) { // wrong in some lane: the six numbers again, from the verified outcome of every lane
            kk = X.kk2[lane];
            live = live_before && (X.ok[lane] != 0);
            gains(X.rdiag2[lane], X.free2[lane] != 0);
            hand_over(i + 1, J.b1, J.b4);
            pair_barrier(); // #1b
        }
        if (JAC_FROM_MASTER && i < T - 1) { // the Jacobian of step i-1, formed by the master during the step before
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    Jn.a[r][c] = X.ja[r * 3 + c][lane];
            Jn.b1 = X.ja[12][lane];
            Jn.b4 = X.ja[13][lane];
        }
        PP_T(t1);
        PP_ADD(1, 0, t0, t1);
        if (pending) { // step i+1: its K, and every entry of its value update
            // (arithmetic on registers may be scheduled across a barrier; what follows must not drift back
            // in front of the hand-over, where the master would wait for it: pin its inputs here)
#pragma unroll
            for (int a = 0; a < NX; ++a) {
                asm volatile("" : "+v"(Kr[a]));
                asm volatile("" : "+v"(T1[a]));
            }
            asm volatile("" : "+v"(kk));
            if (act)
                store5(K, ld, (int64_t)(i + 1) * NX, b, Kr);
            // V_x = Q_x + K^T Q_uu k + K^T Q_u + Q_ux^T k ; V_xx = Q_xx + K^T Q_uu K + K^T Q_ux + Q_ux^T K,
            // symmetrised   (:164-169)
            double Vn[NX * NX];
#pragma unroll
            for (int a = 0; a < NX; ++a) {
                S.Vx[a] = ((Qx[a] + T1[a] * kk) + Kr[a] * Qu) + Qux[a] * kk;
#pragma unroll
                for (int bb = 0; bb < NX; ++bb)
                    Vn[a * NX + bb] = ((Qxx[a * NX + bb] + T1[a] * Kr[bb]) + Kr[a] * Qux[bb]) + Qux[a] * Kr[bb];
            }
#pragma unroll
            for (int a = 0; a < NX; ++a)
#pragma unroll
                for (int bb = a; bb < NX; ++bb) {
                    double v = 0.5 * (Vn[a * NX + bb] + Vn[bb * NX + a]);
                    S.Vxx[a * NX + bb] = v;
                    S.Vxx[bb * NX + a] = v;
                }
#pragma unroll
            for (int a = 0; a < NX; ++a)
                x[a] = xq[a];
            u = uq;
            // (the update itself is wanted BEFORE the next hand-over, in the time the master spends on its guess:
            // pin its results here; the Q-terms below may be scheduled on either side)
#pragma unroll
            for (int a = 0; a < NX; ++a) {
                asm volatile("" : "+v"(S.Vx[a]));
#pragma unroll
                for (int bb = a; bb < NX; ++bb)
                    asm volatile("" : "+v"(S.Vxx[a * NX + bb]));
            }
#pragma unroll
            for (int a = 0; a < NX; ++a)
#pragma unroll
                for (int bb = 0; bb < a; ++bb)
                    S.Vxx[a * NX + bb] = S.Vxx[bb * NX + a];
        }
        {   // operands of step i-1 (for its Jacobian, formed at the end of this step's Q-terms)
            const int iq = i > 0 ? i - 1 : 0;
            load5(xs, ld, (int64_t)iq * NX, b, xq);
            uq = us[(int64_t)iq * ld + b];
        }
        // ---- the part of _compute_Q_terms (:179-228) that does not wait for the boxQP
        C.at(i);
        double lx[NX];
        C.lx(x, lx);
        const double lu = C.lu(u);
        const double(&A)[4][3] = J.a;
        const double* Vx = S.Vx;
        const double* Vxx = S.Vxx;
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
        Qu = lu + fma(J.b4, Vx[4], J.b1 * Vx[1]);
        double M1[NX * NX];
#pragma unroll
        for (int bb = 0; bb < NX; ++bb) {
            M1[0 * NX + bb] = Vxx[0 * NX + bb];
            M1[1 * NX + bb] = dt * Vxx[0 * NX + bb] + Vxx[1 * NX + bb];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double acc = A[0][c] * Vxx[1 * NX + bb];
                acc = fma(A[1][c], Vxx[2 * NX + bb], acc);
                acc = fma(A[2][c], Vxx[3 * NX + bb], acc);
                acc = fma(A[3][c], Vxx[4 * NX + bb], acc);
                M1[(2 + c) * NX + bb] = acc;
            }
        }
#if defined(ILQR_PIN_M1)
        // (with three wavefronts the hand-over of the guess waits for the verifier's first boxQP pass: this much of
        // the Q-terms fits in front of it)
#pragma unroll
        for (int a = 1; a < NX; ++a)
#pragma unroll
            for (int bb = 0; bb < NX; ++bb)
                asm volatile("" : "+v"(M1[a * NX + bb]));
#endif
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
        double M2[NX];
#pragma unroll
        for (int bb = 0; bb < NX; ++bb) {
            double r1 = Vxx[1 * NX + bb] + (bb == 1 ? mu : 0.0);
            double r4 = Vxx[4 * NX + bb] + (bb == 4 ? mu : 0.0);
            M2[bb] = fma(J.b4, r4, J.b1 * r1);
        }
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
        Quu = C.luu() + fma(M2[4], J.b4, M2[1] * J.b1);
#if !defined(ILQR_NO_PIN_QUX)
        // (what the gains wait for after the hand-over of the guess: ready before it)
#pragma unroll
        for (int bb = 0; bb < NX; ++bb)
            asm volatile("" : "+v"(Qux[bb]));
        asm volatile("" : "+v"(Quu));
#endif
#if defined(ILQR_PIN_SIX)
        asm volatile("" : "+v"(Qxx[1 * NX + 1]));
        asm volatile("" : "+v"(Qxx[1 * NX + 4]));
        asm volatile("" : "+v"(Qxx[4 * NX + 1]));
        asm volatile("" : "+v"(Qxx[4 * NX + 4]));
        asm volatile("" : "+v"(Qx[1]));
        asm volatile("" : "+v"(Qx[4]));
        asm volatile("" : "+v"(Qu));
#endif
        // ---- the Jacobian of step i-1 (operands xq, uq; its next state is x), while the master is still in its boxQP
        if (!JAC_FROM_MASTER) {
            cartpole_jac(P, xq, uq, x[2], x[3], J);
            Jn = J;
        }
        PP_T(t2);
        PP_ADD(1, 1, t1, t2);
        pair_barrier(); // #2: the master's guess of k, factor and free mask of step i is in X
        PP_T(t3);
        PP_ADD(1, 2, t2, t3);
        kk = X.kk[lane];
        live_before = live;
        live = live && (X.ok[lane] != 0);
        gains(X.rdiag[lane], X.free_[lane] != 0);
        hand_over(i, Jn.b1, Jn.b4);
        if (JAC_FROM_MASTER)
            J = Jn;
        pending = true;
        PP_T(t4);
        PP_ADD(1, 3, t3, t4);
    }
    pair_barrier(); // the master's verdict on its guess of step 0
    if (X.fix != 0) {
        live = live_before && (X.ok[lane] != 0);
        gains(X.rdiag2[lane], X.free2[lane] != 0);
    }
    if (act) // step 0
        store5(K, ld, 0, b, Kr);
    PP_FLUSH(1);
}

// ------------------------------------------------------------- whole solve

// Diagnostic: when set (ilqr_debug_wave_log), every wave of k_solve records its start
// and end time (100 MHz real-time counter) and the hardware id of the SIMD it ran on.
static __device__ unsigned long long* g_wave_log = nullptr;

struct SolveBuffers {       // batch-minor workspace, two copies of everything that is swapped
    double* xs[2];          // [(T+1)*5, ld]
    double* us[2];          // [T, ld]
    double* k[2];           // [T, ld]
    double* K[2];           // [T*5, ld]
    int32_t* sel_traj;      // [B] which copy holds the final trajectory
    int32_t* sel_gain;      // [B] which copy holds the final gains
    const double* goal_x;   // per-problem goals: [goal_rows_x, ld] and [goal_rows_u, ld], or null (shared goals)
    const double* goal_u;
    int64_t goal_rows_x, goal_rows_u;
};

// Line-search candidates evaluated cooperatively by a wave in one pass.
constexpr int LS_PASS = 8;   // problems per cooperative pass
constexpr int LS_MAXA = 32;  // largest alpha schedule

// iLQR.update_u (ilqr_core.py:241-399): every lane runs its own problem to
// termination; no host involvement per iteration.
//
// Line search (:301-313).  The reference tries the alphas one after the other and
// stops at the first hit; the first alpha is accepted in > 99.9 % of the iterations
// of the cart-pole workloads.  Every lane therefore rolls out alpha_0 for its own
// problem; the few problems that need more hand their remaining candidates to the
// whole wave: each lane takes one (problem, alpha) pair, the costs land in LDS, and
// the owner scans them in the reference's order.  The accept / reject decision and
// every number are those of the serial loop (a rollout is a pure function of
// (problem, alpha)); only the cost changes: a rejected iteration needs ~3 rollout
// times instead of 30, which removes the straggler waves that otherwise set the
// kernel's duration.
// Compiled for one wave per SIMD (up to 512 registers, no spills): a build for two
// waves per SIMD spills ~400 bytes per lane and is slower at every batch size measured
// (158 vs 114 ms at 65536 problems, 227 vs 214 ms at 131072).
//
// PAIR: the launch consists of 128-thread workgroups, wavefront 0 of each the master that runs everything
// below, wavefront 1 its helper in the backward sweeps (see pair_master_sweep).
struct NoPair {
};
template <class Cost, int GROUP>
__global__ __launch_bounds__(256, 1) void k_solve(CartpoleP mp, QRCostP Cp, SolveOptsP O, BoundsP bd, int T,
                                                int64_t B, SolveBuffers W,
                                                double* __restrict__ delta_io,
                                                double* __restrict__ J_out, double* __restrict__ mu_out,
                                                int32_t* __restrict__ iters_out,
                                                int32_t* __restrict__ status_out, ilqr_trace tr)
{
    __shared__ double s_alpha[LS_MAXA];
    __shared__ double s_J[4][LS_PASS][LS_MAXA];
    __shared__ int s_lane[4][LS_PASS], s_cur[4][LS_PASS], s_gcur[4][LS_PASS];
    __shared__ double s_qp_tab[QP_TAB];
    constexpr bool PAIR = GROUP >= 2; // wavefront 0 is the master; 1 the helper; 2 (GROUP == 3) the verifier
    __shared__ typename std::conditional<PAIR, PairX, NoPair>::type s_pair;

    // problems per wavefront (64, 32 or 16): a small batch is spread thinly over all SIMDs, which costs a
    // wave nothing (an instruction takes as long for 16 lanes as for 64) and makes the wave-level slow
    // paths (boxQP fallbacks, the cooperative line search) rarer per wave
    const int lane = threadIdx.x & 63;
    const int lpw = O.lanes_per_wave;
    // PAIR: one group of problems per workgroup.  Groups 2j and 2j+1 share a 64-column workspace tile when
    // lpw = 32; consecutive workgroups go to different XCDs, so the tile-mates are workgroups 16c + r and
    // 16c + r + 8 (the same XCD under round-robin placement) rather than neighbours.
    const int64_t wave_global =
        PAIR ? ((int64_t)(blockIdx.x >> 4) << 4) + ((blockIdx.x & 7) << 1) + ((blockIdx.x >> 3) & 1)
             : ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t gb = wave_global * lpw + lane; // problem index
    const bool valid = lane < lpw && gb < B;
    const int wv = PAIR ? 0 : threadIdx.x >> 6; // wave within the workgroup (each wave only talks to itself)
    if (threadIdx.x < LS_MAXA)
        s_alpha[threadIdx.x] = O.alphas[threadIdx.x];
    qp_fill_step_table(s_qp_tab, O.qp.step_dec); // ends with the workgroup barrier

    unsigned long long* wave_log = g_wave_log;
    if (wave_log != nullptr && lane == 0 && valid) {
        unsigned hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        wave_log[wave_global * 4 + 0] = __builtin_amdgcn_s_memrealtime();
        wave_log[wave_global * 4 + 2] = hw_id;
        wave_log[wave_global * 4 + 3] = xcc;
    }
    // the workspace is wave-tiled: this wave's rows are contiguous, leading dimension 64.
    // (Named pointers and selects, not arrays indexed by a lane-varying 0/1: a
    // dynamically indexed pointer array would live in scratch memory.)
    const int64_t tile = (wave_global * lpw) >> 6; // wave-uniform: the helper lanes of the line search use it too
    double* const xs0 = W.xs[0] + tile * (int64_t)(T + 1) * NX * 64;
    double* const xs1 = W.xs[1] + tile * (int64_t)(T + 1) * NX * 64;
    double* const us0 = W.us[0] + tile * (int64_t)T * 64;
    double* const us1 = W.us[1] + tile * (int64_t)T * 64;
    double* const k0 = W.k[0] + tile * (int64_t)T * 64;
    double* const k1 = W.k[1] + tile * (int64_t)T * 64;
    double* const K0 = W.K[0] + tile * (int64_t)T * NX * 64;
    double* const K1 = W.K[1] + tile * (int64_t)T * NX * 64;
#define XS(i) ((i) ? xs1 : xs0)
#define US(i) ((i) ? us1 : us0)
#define KF(i) ((i) ? k1 : k0)
#define KG(i) ((i) ? K1 : K0)
    const int64_t ld = 64;
    const int64_t b = gb & 63; // column of the problem in its tile
    // lanes without a problem take part in the two-wavefront sweeps with a column of their own group, so that
    // their (discarded) loads touch no memory the group does not read anyway
    const int64_t bq = valid ? b : ((wave_global * lpw) & 63);
    CartpoleC P = cartpole_consts(mp);
    Cost C(Cp);
    if (W.goal_x != nullptr) // every problem has its own goal or path: rows of this wavefront's tile
        C.bind(W.goal_x + tile * W.goal_rows_x * 64, W.goal_u + tile * W.goal_rows_u * 64, ld);
    C.column(b);
    QPOptsP QO = qp_opts_in_vgprs(O.qp);
    if constexpr (PAIR) {
        if ((threadIdx.x >> 6) == 1) { // the helper: one backward sweep per command of the master
            for (;;) {
                __syncthreads(); // the master's command, and its stores to the trajectory (global memory)
                const int cmd = s_pair.cmd;
                if (cmd == 0)
                    break;
                const bool act = s_pair.act[lane] != 0;
                const int pc = s_pair.cur[lane], pg = s_pair.gnew[lane];
                if (cmd == 2) { // the angle chain of the master's rollout
                    const int sn = GROUP == 3 ? s_pair.spec_n : 0;
                    forward_rollout_helper(P, T, ld, bq, s_pair, lane, XS(pc));
                    if (sn > 0)
                        __syncthreads(); // (the verifier's costs reach the master)
                    continue;
                }
                pair_helper_sweep<Cost, GROUP == 3>(P, C, T, ld, bq, s_pair, lane, act, XS(pc), US(pc), s_pair.mu[lane],
                                                    KG(pg));
                __syncthreads(); // K is stored: the master's rollouts may read it
            }
            return;
        }
        if (GROUP == 3 && (threadIdx.x >> 6) == 2) { // the verifier
            for (;;) {
                __syncthreads();
                const int cmd = s_pair.cmd;
                if (cmd == 0)
                    break;
                if (cmd == 2) { // a rollout shared by the other two
                    const int sn = s_pair.spec_n;
                    if (sn > 0) {
                        // ... beside which this wavefront tries the further step sizes of the problems that needed
                        // them in the iteration before (lane = problem and step size), attending the hand-overs
                        const int n_more = O.n_alpha - 1;
                        const bool on = lane < sn * n_more;
                        const int fi = on ? lane / n_more : 0, jj = on ? 1 + lane % n_more : 1;
                        const int pc = s_pair.spec_cur[fi], pg = s_pair.spec_gcur[fi];
                        const double Jc = forward_rollout<false, Cost, true>(P, C, T, ld, (int64_t)s_pair.spec_col[fi], XS(pc),
                                                                             US(pc), KF(pg), KG(pg), s_alpha[jj], nullptr,
                                                                             nullptr);
                        if (on)
                            s_pair.specJ[fi][jj] = Jc;
                        __syncthreads();
                    } else { // ... only their hand-overs to attend
                        for (int t = 0; t < T; ++t)
                            pair_barrier();
                    }
                    continue;
                }
                const bool act = s_pair.act[lane] != 0;
                const int pc = s_pair.cur[lane], pg = s_pair.gnew[lane];
                pair_verifier_sweep(QO, s_qp_tab, bd, T, ld, bq, s_pair, lane, act, US(pc), KF(pg));
                __syncthreads(); // k is stored
            }
            return;
        }
    }
    int cur = 0;  // trajectory copy in use
    int gcur = 0; // gain copy in use (holds the solver's k_/K_ on entry)
    double mu = 1.0, delta = 2.0, J = 0.0;
    if (valid) {
        mu = O.mu_from_array ? mu_out[gb] : 1.0; // reset per call (:264); delta_ persists (:62)
        delta = delta_io[gb];
        // initial rollout and cost (:269-271); us_init is in W.us[0], x0 in rows 0..4 of W.xs[0]
        double x[NX], xn[NX];
        load5(xs0, ld, 0, b, x);
        for (int t = 0; t < T; ++t) {
            double u = us0[(int64_t)t * ld + b];
            if (t > 0)
                store5(xs0, ld, (int64_t)t * NX, b, x);
            C.at(t);
            J = J + C.l(x, u);
            cartpole_f(P, x, u, xn);
#pragma unroll
            for (int i = 0; i < NX; ++i)
                x[i] = xn[i];
        }
        store5(xs0, ld, (int64_t)T * NX, b, x);
        C.at_terminal(T);
        J = J + C.l_terminal(x);
    }

    double dJ1 = 0.0, dJ2 = 0.0, actual = 0.0;
    if (valid && O.mu_from_array && O.dJ_carry != nullptr) { // a continued solve: the sums of its last backward pass
        dJ1 = O.dJ_carry[gb];
        dJ2 = O.dJ_carry[B + gb];
    }
    int converged = 0, status = ILQR_RAN_OUT;
    int iters = 0;
    bool past_first = false; // the line search of the iteration before went past the first step size
    bool running = valid && O.max_iter > 0;
    const int n_more = O.n_alpha - 1; // candidates after alpha_0

    PP_DECL;
    for (int it = 0; it < O.max_iter; ++it) {
        if (__ballot(running) == 0ull)
            break;
        PP_T(ti0);
        BwAcc S;
        int stopped_at = -1;
        S.qp_iter = S.qp_backtrack = S.n_clamped = 0;
        int accepted = 0, j = 0;
        double Jnew = 0.0;
        const int nxt = cur ^ 1;
        bool need = false; // the serial loop would go on to alpha_1
        if constexpr (PAIR) {
            // ---- backward pass shared with the helper wavefront; the LinAlgError retry loop (:287-299) as a
            // wave-uniform loop over the lanes that still want a sweep
            bool want = running;
            while (__ballot(want) != 0ull) {
                const int gnew = gcur ^ 1;
                s_pair.act[lane] = want ? 1 : 0;
                s_pair.mu[lane] = mu;
                s_pair.cur[lane] = cur;
                s_pair.gnew[lane] = gnew;
                if (lane == 0)
                    s_pair.cmd = 1;
                __syncthreads();
                BwAcc Sn;
                int sa;
                int rc = pair_master_sweep<Cost, GROUP == 2>(C, QO, s_qp_tab, bd, T, ld, bq, s_pair, lane, want, US(cur),
                                                             mu, KF(gnew), Sn, sa, P, XS(cur));
                __syncthreads(); // the helper's K is stored
                if (want) {
                    S = Sn;
                    stopped_at = sa;
                    if (rc != BW_LINALG) {
                        gcur = gnew;
                        dJ1 = Sn.dJ1;
                        dJ2 = Sn.dJ2;
                        want = false;
                    } else {
                        delta = fmax(delta, 1.0) * O.delta0;
                        mu = fmax(O.mu_min, mu * delta);
                        if (O.mu_max != 0.0 && mu >= O.mu_max)
                            want = false;
                    }
                }
            }
        }
        PP_T(ti1);
        PP_ADD(0, 5, ti0, ti1);
        int spec_slot = -1, spec_n = 0; // (three wavefronts) this lane's place among the problems the verifier tries ahead
        if constexpr (PAIR) {
            if (O.n_alpha > 0) { // the first rollout is shared with the helper wavefront
                s_pair.cur[lane] = cur;
                if constexpr (GROUP == 3) {
                    const bool want_spec = past_first && running && n_more > 0 && n_more * 2 <= 64;
                    const unsigned long long pm = __ballot(want_spec);
                    if (pm != 0ull) {
                        const int r = __popcll(pm & ((1ull << lane) - 1ull));
                        if (want_spec && r < 2) {
                            spec_slot = r;
                            s_pair.spec_col[r] = (int)b;
                            s_pair.spec_cur[r] = cur;
                            s_pair.spec_gcur[r] = gcur;
                        }
                        spec_n = __popcll(pm) < 2 ? __popcll(pm) : 2;
                    }
                    if (lane == 0)
                        s_pair.spec_n = spec_n;
                }
                if (lane == 0)
                    s_pair.cmd = 2;
                __syncthreads();
            }
        }
        if (running) {
            const double* xs = XS(cur);
            const double* us = US(cur);
            // ---- backward pass, with the LinAlgError retry loop (:287-299)
            bool done = PAIR;
            while (!done) {
                int gnew = gcur ^ 1;
                int rc = backward_sweep<true, false>(P, C, QO, s_qp_tab, bd, T, ld, b, xs, us, mu, KF(gnew), KG(gnew),
                                                     S, stopped_at, nullptr, nullptr);
                if (rc != BW_LINALG) {
                    gcur = gnew;
                    dJ1 = S.dJ1;
                    dJ2 = S.dJ2;
                    done = true;
                } else {
                    delta = fmax(delta, 1.0) * O.delta0;
                    mu = fmax(O.mu_min, mu * delta);
                    if (O.mu_max != 0.0 && mu >= O.mu_max)
                        break;
                }
            }
            // ---- line search, alpha_0 (:301-313)
            if (O.n_alpha > 0) {
                double alpha = s_alpha[0];
                if constexpr (PAIR)
                    Jnew = forward_rollout_master(P, C, T, ld, b, s_pair, lane, xs, us, KF(gcur), KG(gcur), alpha, XS(nxt),
                                                  US(nxt));
                else
                    Jnew = forward_rollout<true>(P, C, T, ld, b, xs, us, KF(gcur), KG(gcur), alpha, XS(nxt), US(nxt));
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
        PP_T(ti2);
        PP_ADD(0, 6, ti1, ti2);
        bool redo = false; // accepted at some alpha_j, j >= 1: its trajectory still has to be written
        // the reference's loop from alpha_1 on, on tabulated costs
        auto scan_costs = [&](const double* row) {
            for (j = 1; j < O.n_alpha; ++j) {
                double alpha = s_alpha[j];
                Jnew = row[j];
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
        };
        past_first = need;
        if constexpr (GROUP == 3) {
            if (spec_n > 0) { // the verifier has tried the further step sizes of up to two problems already
                __syncthreads();
                if (need && spec_slot >= 0)
                    scan_costs(s_pair.specJ[spec_slot]);
                if (lane == 0)
                    s_pair.spec_n = 0; // (a re-run below is a shared rollout without this)
            }
        }
        // ---- remaining alphas of the problems that need them, LS_PASS problems per pass
        // (helper lanes read gains another lane of this wave has just stored)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        unsigned long long mask = __ballot(need);
        while (mask != 0ull) {
            int rank = __popcll(mask & ((1ull << lane) - 1ull));
            bool mine = need && rank < LS_PASS;
            if (mine) {
                s_lane[wv][rank] = (int)b;
                s_cur[wv][rank] = cur;
                s_gcur[wv][rank] = gcur;
            }
            __builtin_amdgcn_wave_barrier();
            int np = __popcll(mask);
            np = np < LS_PASS ? np : LS_PASS;
            int nc = np * n_more;
            for (int q0 = 0; q0 < nc; q0 += 64) {
                int q = q0 + lane;
                if (q < nc) {
                    int fi = q / n_more, jj = 1 + q % n_more;
                    int64_t pb = s_lane[wv][fi];
                    int pc = s_cur[wv][fi], pg = s_gcur[wv][fi];
                    s_J[wv][fi][jj] = forward_rollout<false>(P, C, T, ld, pb, XS(pc), US(pc), KF(pg), KG(pg),
                                                              s_alpha[jj], nullptr, nullptr);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (mine)
                scan_costs(s_J[wv][rank]);
            __builtin_amdgcn_wave_barrier();
            mask = __ballot(need);
        }
        // the accepted candidate's states and controls (same arithmetic, same bits)
        if constexpr (PAIR) {
            if (__ballot(redo) != 0ull) { // shared with the helper wavefront like the first rollout
                s_pair.cur[lane] = cur;
                if (lane == 0)
                    s_pair.cmd = 2;
                __syncthreads();
                if (redo)
                    Jnew = forward_rollout_master(P, C, T, ld, b, s_pair, lane, XS(cur), US(cur), KF(gcur), KG(gcur),
                                                  s_alpha[j], XS(nxt), US(nxt));
            }
        } else if (redo) {
            Jnew = forward_rollout<true>(P, C, T, ld, b, XS(cur), US(cur), KF(gcur), KG(gcur), s_alpha[j], XS(nxt),
                                         US(nxt));
        }
        // ---- regularisation schedule and termination (:315-339)
        if (running) {
            bool stop = false;
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
            int64_t to = (int64_t)it * B + gb;
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
        PP_T(ti3);
        PP_ADD(0, 7, ti2, ti3);
    }
    PP_FLUSH(0);
    if (valid) {
        delta_io[gb] = delta;
        J_out[gb] = J;
        mu_out[gb] = mu;
        iters_out[gb] = iters;
        status_out[gb] = status;
        W.sel_traj[gb] = cur;
        W.sel_gain[gb] = gcur;
        if (O.mu_from_array && O.dJ_carry != nullptr) { // only a solve that is being continued carries the sums
            O.dJ_carry[gb] = dJ1;
            O.dJ_carry[B + gb] = dJ2;
        }
    }
    if (wave_log != nullptr && lane == 0 && valid)
        wave_log[wave_global * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    if constexpr (PAIR) { // release the helper
        if (lane == 0)
            s_pair.cmd = 0;
        __syncthreads();
    }
#undef XS
#undef US
#undef KF
#undef KG
}

// ------------------------------------------------------- stand-alone boxQP

// boxQP (ilqr_boxQP.py:9-186) for m <= 8, one QP per lane, ref layout (boxqp_dense).
__global__ __launch_bounds__(WAVE) void k_boxqp(QPOptsP O, int64_t B, int m, const double* __restrict__ Hg,
                                                const double* __restrict__ gg,
                                                const double* __restrict__ lowerg,
                                                const double* __restrict__ upperg,
                                                const double* __restrict__ x0g, double* __restrict__ xg_out,
                                                int32_t* __restrict__ result_out, double* __restrict__ Hff_out,
                                                int32_t* __restrict__ hd_out, int32_t* __restrict__ free_out,
                                                int32_t* __restrict__ cnt_out)
{
    int64_t q = (int64_t)blockIdx.x * WAVE + threadIdx.x;
    if (q >= B)
        return;
    double H[QM * QM], g[QM], lower[QM], upper[QM], x[QM], Hff[QM * QM];
    int freev[QM];
    for (int i = 0; i < m * m; ++i)
        H[i] = Hg[q * m * m + i];
    for (int i = 0; i < m; ++i) {
        g[i] = gg[q * m + i];
        lower[i] = lowerg[q * m + i];
        upper[i] = upperg[q * m + i];
        x[i] = x0g ? x0g[q * m + i] : 0.0;
    }
    QPDenseOut r;
    boxqp_dense(O, m, H, g, lower, upper, x0g != nullptr, x, Hff, freev, r);
    for (int i = 0; i < m; ++i) {
        xg_out[q * m + i] = x[i];
        free_out[q * m + i] = freev[i];
    }
    for (int i = 0; i < m * m; ++i)
        Hff_out[q * m * m + i] = Hff[i];
    result_out[q] = r.linalg ? -99 : r.result; // -99: the reference would raise LinAlgError
    hd_out[q] = r.hd;
    cnt_out[q * 3 + 0] = r.n_iter;
    cnt_out[q * 3 + 1] = r.n_factor;
    cnt_out[q * 3 + 2] = r.n_bt;
}

// boxQP for m = 1 through the same in-register routine the backward pass uses.
__global__ __launch_bounds__(256) void k_boxqp1(QPOptsP O, int64_t B, const double* __restrict__ H,
                                                const double* __restrict__ g, const double* __restrict__ lower,
                                                const double* __restrict__ upper, const double* __restrict__ x0,
                                                double* __restrict__ x_out, int32_t* __restrict__ result_out,
                                                double* __restrict__ Hff_out, int32_t* __restrict__ hd_out,
                                                int32_t* __restrict__ free_out, int32_t* __restrict__ cnt_out)
{
    __shared__ double s_qp_tab[QP_TAB];
    qp_fill_step_table(s_qp_tab, O.step_dec);
    int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= B)
        return;
    QP1Out out;
    double lo = lower[q], hi = upper[q];
    double xs = x0 ? x0[q] : (lo + hi) / 2;
    double sq = sqrt(H[q]);
    boxqp1(O, s_qp_tab, H[q], sq, 1.0 / sq, g[q], lo, hi, xs, out);
    if (!(fabs(H[q]) < __builtin_inf()) && O.max_iter > 0) {
        // non-finite H: the first pass either finds everything clamped (:92) or
        // np.linalg.eigvals raises (:103); boxqp1 is written for the finite H the
        // backward pass guarantees, so that case is decided here
        double xq = clamp1(lo, hi, xs);
        double grad = g[q] + H[q] * xq;
        bool cl = ((xq == lo) && (grad > 0.0)) || ((xq == hi) && (grad < 0.0));
        out.x = xq;
        out.result = cl ? 6 : -99;
        out.free_ = !cl;
        out.hff = 1.0;
        out.n_iter = out.n_factor = out.n_backtrack = 0;
    }
    x_out[q] = out.x;
    result_out[q] = out.result;
    Hff_out[q] = out.hff;
    hd_out[q] = 1;
    free_out[q] = out.free_ ? 1 : 0;
    cnt_out[q * 3 + 0] = out.n_iter;
    cnt_out[q * 3 + 1] = out.n_factor;
    cnt_out[q * 3 + 2] = out.n_backtrack;
}

// -------------------------------------------------------------- plant step

// CartPoleContinuousEnv._state_eq, ilqr/cartpole_gym.py:60-72
struct PlantP {
    double g, mc, mp, len, tau;
};

__device__ __forceinline__ void plant_step_one(const PlantP& p, const double (&s)[4], double force, double (&out)[4])
{
    double M = p.mp + p.mc, pml = p.mp * p.len;
    double x = s[0], xd = s[1], th = s[2], thd = s[3];
    double sn, cs;
    det_sincos(th, sn, cs);
    double temp = (force + pml * thd * thd * sn) / M;
    double thacc = (p.g * sn - cs * temp) / (p.len * (4.0 / 3.0 - p.mp * cs * cs / M));
    double xacc = temp - pml * thacc * cs / M;
    out[0] = x + p.tau * xd;
    out[1] = xd + p.tau * xacc;
    out[2] = th + p.tau * thd;
    out[3] = thd + p.tau * thacc;
}

__global__ __launch_bounds__(256) void k_plant_step(PlantP p, int64_t B, const double* __restrict__ st,
                                                    const double* __restrict__ u, double* __restrict__ out)
{
    int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B)
        return;
    double s[4] = {st[b * 4 + 0], st[b * 4 + 1], st[b * 4 + 2], st[b * 4 + 3]}, o[4];
    plant_step_one(p, s, u[b], o);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        out[b * 4 + i] = o[i];
}

// Everything between two solves of a closed loop (examples/cartpole_example_mpc.py:66-79 around
// ilqr_core.py:451-465) in one launch: apply the first control of the solved sequence to the plant, augment
// the new state for the next solve (cartpole.py:97-116), and seed the next solve with the sequence moved one
// step along, its last control repeated (ilqr_core.py:463-465).  Same device functions, hence the same bits,
// as the separate kernels.
__global__ __launch_bounds__(256) void k_mpc_advance(PlantP p, int64_t B, int64_t N, double* __restrict__ st,
                                                     const double* __restrict__ us_solved,
                                                     double* __restrict__ us_seed, double* __restrict__ u_applied,
                                                     double* __restrict__ x0_aug)
{
    int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B)
        return;
    double s[4] = {st[b * 4 + 0], st[b * 4 + 1], st[b * 4 + 2], st[b * 4 + 3]}, o[4];
    double u0 = us_solved[b * N];
    plant_step_one(p, s, u0, o);
    u_applied[b] = u0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        st[b * 4 + i] = o[i];
    double sn, cs;
    det_sincos(o[2], sn, cs);
    x0_aug[b * 5 + 0] = o[0];
    x0_aug[b * 5 + 1] = o[1];
    x0_aug[b * 5 + 2] = sn;
    x0_aug[b * 5 + 3] = cs;
    x0_aug[b * 5 + 4] = o[3];
    for (int64_t t = 0; t < N; ++t)
        us_seed[b * N + t] = us_solved[b * N + (t + 1 < N ? t + 1 : N - 1)];
}

__global__ __launch_bounds__(256) void k_detmath(int64_t n, const double* a, const double* bb, double* s,
                                                 double* c, double* t, double* sq, double* dv)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    double sn, cs;
    det_sincos(a[i], sn, cs);
    s[i] = sn;
    c[i] = cs;
    t[i] = det_atan2(a[i], bb[i]);
    sq[i] = sqrt(fabs(a[i]));
    dv[i] = a[i] / bb[i];
}

// --------------------------------------------------- host-side conversions

int check_model(const ilqr_model* m)
{
    if (!m)
        return ILQR_ERR_ARG;
    if (m->model != ILQR_MODEL_CARTPOLE || m->n != 5 || m->m != 1 || (m->cost_kind != ILQR_COST_QR && m->cost_kind != ILQR_COST_PATH_QR) || m->control_limit != 1)
        return ILQR_ERR_UNSUPPORTED;
    if (m->T < 1 || !m->Q || !m->R || !m->lower || !m->upper)
        return ILQR_ERR_ARG;
    if (m->cost_kind == ILQR_COST_PATH_QR && (!m->x_goal || !m->u_goal)) // the paths are mandatory
        return ILQR_ERR_ARG;
    if ((m->goal_stride_x != 0) != (m->goal_stride_u != 0))
        return ILQR_ERR_ARG;
    if (m->goal_stride_x != 0) { // one goal or path per problem, device arrays
        const bool path = m->cost_kind == ILQR_COST_PATH_QR;
        if (!m->x_goal || !m->u_goal || m->goal_stride_x < (path ? ((int64_t)m->T + 1) * 5 : 5) ||
            m->goal_stride_u < (path ? (int64_t)m->T : 1))
            return ILQR_ERR_ARG;
    }
    return ILQR_OK;
}

inline bool per_problem_goals(const ilqr_model* m) { return m->goal_stride_x != 0; }
// the stage entry points evaluate shared goals only
inline int check_stage_model(const ilqr_model* m)
{
    int rc = check_model(m);
    if (rc == ILQR_OK && per_problem_goals(m))
        rc = ILQR_ERR_UNSUPPORTED;
    return rc;
}

void fill_model(const ilqr_model* m, CartpoleP& mp, QRCostP& C, BoundsP& bd)
{
    mp.dt = m->dt;
    mp.mc = m->mc;
    mp.mp = m->mp;
    mp.l = m->l;
    mp.g = m->g;
    const double* Qt = m->Q_terminal ? m->Q_terminal : m->Q; // QRCost default, ilqr_cost.py:504-505
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b) {
            C.Q[a * 5 + b] = m->Q[a * 5 + b];
            C.Qt[a * 5 + b] = Qt[a * 5 + b];
            C.QQ[a * 5 + b] = m->Q[a * 5 + b] + m->Q[b * 5 + a];
            C.QQt[a * 5 + b] = Qt[a * 5 + b] + Qt[b * 5 + a];
        }
    C.R = m->R[0];
    C.RR = m->R[0] + m->R[0];
    C.goal_t_x = C.goal_t_u = 0;
    C.vscale = 1.0;
    if (per_problem_goals(m)) { // device arrays, packed into the solve's workspace; the path evaluator reads them there
        const bool path = m->cost_kind == ILQR_COST_PATH_QR;
        for (int a = 0; a < 5; ++a)
            C.xg[a] = 0.0;
        C.ug = 0.0;
        C.x_path = m->x_goal; // (marks the evaluator; the kernels do not read these two pointers)
        C.u_path = m->u_goal;
        C.goal_t_x = path ? 5 : 0;
        C.goal_t_u = path ? 1 : 0;
        C.vscale = path ? 1.0 : 0.5; // QRCost values carry the 1/2 (ilqr_cost.py:547, :553)
    } else if (m->cost_kind == ILQR_COST_PATH_QR) { // device arrays, read by the kernels
        for (int a = 0; a < 5; ++a)
            C.xg[a] = 0.0;
        C.ug = 0.0;
        C.x_path = m->x_goal;
        C.u_path = m->u_goal;
    } else {
        for (int a = 0; a < 5; ++a)
            C.xg[a] = m->x_goal ? m->x_goal[a] : 0.0;
        C.ug = m->u_goal ? m->u_goal[0] : 0.0;
        C.x_path = nullptr;
        C.u_path = nullptr;
    }
    bd.lo = m->lower[0];
    bd.hi = m->upper[0];
}

bool cost_is_diagonal(const QRCostP& C)
{
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b)
            if (a != b && (C.Q[a * 5 + b] != 0.0 || C.Qt[a * 5 + b] != 0.0))
                return false;
    return true;
}

// ILQR_FORCE_DENSE_COST=1 routes diagonal costs through the dense evaluator too
// (used by the tests to show that both give the same bits).
bool use_diag(const QRCostP& C)
{
    const char* e = std::getenv("ILQR_FORCE_DENSE_COST");
    if (e && e[0] == '1')
        return false;
    return cost_is_diagonal(C);
}

// Which cost evaluator a launch instantiates.
enum { EVAL_DIAG = 0, EVAL_DENSE = 1, EVAL_PATH = 2 };
int cost_eval(const QRCostP& C)
{
    if (C.x_path != nullptr)
        return EVAL_PATH;
    return use_diag(C) ? EVAL_DIAG : EVAL_DENSE;
}

int fill_qp(const ilqr_opts* o, QPOptsP& q)
{
    if (!o || o->qp_max_iter < 0 || !(o->qp_step_dec > 0.0) || !(o->qp_step_dec < 1.0) ||
        !(o->qp_min_step > 0.0))
        return ILQR_ERR_ARG;
    q.max_iter = o->qp_max_iter;
    q.min_grad = o->qp_min_grad;
    q.min_rel = o->qp_min_rel_improve;
    q.step_dec = o->qp_step_dec;
    q.min_step = o->qp_min_step;
    q.armijo = o->qp_armijo;
    // number of `step *= stepDec` after which step < minStep (ilqr_boxQP.py:151-159),
    // with the same rounded multiplications the loop performs
    volatile double step = 1.0;
    int n = 0;
    do {
        step = step * o->qp_step_dec;
        ++n;
    } while (!(step < o->qp_min_step) && n < 1000000);
    q.n_exhaust = n;
    q.step_last = step;
    q.tab_n = n < QP_TAB ? n : QP_TAB;
    q.inv_log2_dec = (float)(1.0 / std::log2(o->qp_step_dec));
    return ILQR_OK;
}

int fill_opts(const ilqr_opts* o, SolveOptsP& s)
{
    if (!o || !o->alphas || o->n_alpha < 0 || o->n_alpha > LS_MAXA || o->max_iter < 0)
        return ILQR_ERR_ARG;
    s.max_iter = o->max_iter;
    s.n_alpha = o->n_alpha;
    s.mu_from_array = (o->mu_init >= 0.0) ? 1 : 0;
    s.lanes_per_wave = WAVE;
    s.tol = o->tol;
    s.scale = o->scale;
    s.mu_min = o->mu_min;
    s.mu_max = o->mu_max;
    s.delta0 = o->delta0;
    s.dJ_carry = o->dJ_carry;
    for (int i = 0; i < 32; ++i)
        s.alphas[i] = i < o->n_alpha ? o->alphas[i] : 0.0;
    return fill_qp(o, s.qp);
}

inline dim3 tile_grid(int64_t B, int64_t R) { return dim3((unsigned)((B + 63) / 64), (unsigned)((R + 63) / 64)); }

} // namespace

// ==================================================================== C ABI

extern "C" {

int ilqr_version(void) { return 102; }

const char* ilqr_status_string(int s)
{
    switch (s) {
    case ILQR_OK: return "ok";
    case ILQR_ERR_ARG: return "invalid argument";
    case ILQR_ERR_UNSUPPORTED: return "model, dimension or option not supported";
    case ILQR_ERR_HIP: return "HIP runtime error";
    case ILQR_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
    }
}

int64_t ilqr_bm_ld(int64_t B) { return (B + 63) / 64 * 64; }

int ilqr_pack_bm(const double* src, double* dst, int64_t B, int64_t R, int64_t ld, void* stream)
{
    if (!src || !dst || B < 1 || R < 1 || ld < B)
        return ILQR_ERR_ARG;
    hipLaunchKernelGGL(k_pack, tile_grid(B, R), dim3(256), 0, as_stream(stream), src, dst, B, R, ld);
    return hip_status(hipGetLastError());
}

int ilqr_unpack_bm(const double* src, double* dst, int64_t B, int64_t R, int64_t ld, void* stream)
{
    if (!src || !dst || B < 1 || R < 1 || ld < B)
        return ILQR_ERR_ARG;
    hipLaunchKernelGGL(k_unpack, tile_grid(B, R), dim3(256), 0, as_stream(stream), src, src,
                       (const int32_t*)nullptr, dst, B, R, ld);
    return hip_status(hipGetLastError());
}

int ilqr_rollout_cost_bm(const ilqr_model* model, int64_t B, int64_t ld, const double* x0,
                         const double* us, double* xs, double* J, void* stream)
{
    int rc = check_stage_model(model);
    if (rc != ILQR_OK)
        return rc;
    if (!x0 || !us || !xs || !J || B < 1 || ld < B)
        return ILQR_ERR_ARG;
    CartpoleP mp;
    QRCostP C;
    BoundsP bd;
    fill_model(model, mp, C, bd);
#define ILQR_LAUNCH_RC(COST)                                                                              \
    hipLaunchKernelGGL(k_rollout_cost<COST>, dim3((unsigned)((B + WAVE - 1) / WAVE)), dim3(WAVE), 0,         \
                       as_stream(stream), mp, C, model->T, B, ld, x0, us, xs, J)
    switch (cost_eval(C)) {
    case EVAL_DIAG: ILQR_LAUNCH_RC(DiagQR); break;
    case EVAL_DENSE: ILQR_LAUNCH_RC(DenseQR); break;
    default: ILQR_LAUNCH_RC(PathQR); break;
    }
#undef ILQR_LAUNCH_RC
    return hip_status(hipGetLastError());
}

int ilqr_deriv_bm(const ilqr_model* model, int64_t B, int64_t ld, const double* xs, const double* us,
                  const ilqr_deriv_bm_t* out, void* stream)
{
    int rc = check_stage_model(model);
    if (rc != ILQR_OK)
        return rc;
    if (!xs || !us || !out || B < 1 || ld < B || !out->fx || !out->fu || !out->lx || !out->lu ||
        !out->lxx || !out->luu || !out->lux || !out->lxT || !out->lxxT)
        return ILQR_ERR_ARG;
    CartpoleP mp;
    QRCostP C;
    BoundsP bd;
    fill_model(model, mp, C, bd);
    if (C.x_path != nullptr)
        hipLaunchKernelGGL(k_deriv<PathQR>, dim3((unsigned)((B + 511) / 512), (unsigned)(model->T + 1)), dim3(256),
                           0, as_stream(stream), mp, C, model->T, B, ld, xs, us, *out);
    else
        hipLaunchKernelGGL(k_deriv<DenseQR>, dim3((unsigned)((B + 511) / 512), (unsigned)(model->T + 1)), dim3(256),
                           0, as_stream(stream), mp, C, model->T, B, ld, xs, us, *out);
    return hip_status(hipGetLastError());
}

int ilqr_backward_bm(const ilqr_model* model, const ilqr_opts* opts, int64_t B, int64_t ld,
                     const double* xs, const double* us, const double* mu,
                     const ilqr_deriv_bm_t* deriv, double* k, double* K, double* dJ, int32_t* info,
                     int32_t* qp_result_bm, int32_t* free_bm, void* stream)
{
    int rc = check_stage_model(model);
    if (rc != ILQR_OK)
        return rc;
    if (!us || !mu || !k || !K || !dJ || !info || B < 1 || ld < B || (deriv == nullptr && !xs))
        return ILQR_ERR_ARG;
    if ((qp_result_bm == nullptr) != (free_bm == nullptr))
        return ILQR_ERR_ARG;
    if (deriv != nullptr) {
        if (!deriv->fx || !deriv->fu || !deriv->lx || !deriv->lu || !deriv->lxx || !deriv->luu || !deriv->lux ||
            !deriv->lxT || !deriv->lxxT)
            return ILQR_ERR_ARG;
        QPOptsP qs;
        rc = fill_qp(opts, qs);
        if (rc != ILQR_OK)
            return rc;
        BoundsP bds;
        bds.lo = model->lower[0];
        bds.hi = model->upper[0];
        dim3 grid_s((unsigned)((B + WAVE - 1) / WAVE));
        if (qp_result_bm != nullptr)
            hipLaunchKernelGGL(k_backward_staged<true>, grid_s, dim3(WAVE), 0, as_stream(stream), qs, bds, model->T, B,
                               ld, us, mu, *deriv, k, K, dJ, info, qp_result_bm, free_bm);
        else
            hipLaunchKernelGGL(k_backward_staged<false>, grid_s, dim3(WAVE), 0, as_stream(stream), qs, bds, model->T, B,
                               ld, us, mu, *deriv, k, K, dJ, info, qp_result_bm, free_bm);
        return hip_status(hipGetLastError());
    }
    CartpoleP mp;
    QRCostP C;
    BoundsP bd;
    QPOptsP q;
    fill_model(model, mp, C, bd);
    rc = fill_qp(opts, q);
    if (rc != ILQR_OK)
        return rc;
    if ((qp_result_bm == nullptr) != (free_bm == nullptr))
        return ILQR_ERR_ARG;
    dim3 grid((unsigned)((B + WAVE - 1) / WAVE));
    bool rec = qp_result_bm != nullptr;
    int ev = cost_eval(C);
#define ILQR_LAUNCH_BW(COST, REC)                                                                            \
    hipLaunchKernelGGL((k_backward<COST, REC>), grid, dim3(WAVE), 0, as_stream(stream), mp, C, q, bd, model->T, \
                       B, ld, xs, us, mu, k, K, dJ, info, qp_result_bm, free_bm)
    if (ev == EVAL_DIAG && rec)
        ILQR_LAUNCH_BW(DiagQR, true);
    else if (ev == EVAL_DIAG)
        ILQR_LAUNCH_BW(DiagQR, false);
    else if (ev == EVAL_DENSE && rec)
        ILQR_LAUNCH_BW(DenseQR, true);
    else if (ev == EVAL_DENSE)
        ILQR_LAUNCH_BW(DenseQR, false);
    else if (rec)
        ILQR_LAUNCH_BW(PathQR, true);
    else
        ILQR_LAUNCH_BW(PathQR, false);
#undef ILQR_LAUNCH_BW
    return hip_status(hipGetLastError());
}

int ilqr_forward_ls_bm(const ilqr_model* model, int64_t B, int64_t ld, int32_t n_alpha,
                       const double* alphas, const double* xs, const double* us, const double* k,
                       const double* K, double* J_new, double* xs_new, double* us_new, void* stream)
{
    int rc = check_stage_model(model);
    if (rc != ILQR_OK)
        return rc;
    if (!alphas || n_alpha < 1 || n_alpha > 32 || !xs || !us || !k || !K || !J_new || B < 1 || ld < B ||
        ((xs_new == nullptr) != (us_new == nullptr)))
        return ILQR_ERR_ARG;
    CartpoleP mp;
    QRCostP C;
    BoundsP bd;
    fill_model(model, mp, C, bd);
    AlphasP al;
    for (int i = 0; i < 32; ++i)
        al.a[i] = i < n_alpha ? alphas[i] : 0.0;
    dim3 grid((unsigned)((B + WAVE - 1) / WAVE), (unsigned)n_alpha);
    bool st = xs_new != nullptr;
    int ev = cost_eval(C);
#define ILQR_LAUNCH_FW(COST, ST)                                                                              \
    hipLaunchKernelGGL((k_forward_ls<COST, ST>), grid, dim3(WAVE), 0, as_stream(stream), mp, C, al, model->T, B, \
                       ld, xs, us, k, K, J_new, xs_new, us_new)
    if (ev == EVAL_DIAG && st)
        ILQR_LAUNCH_FW(DiagQR, true);
    else if (ev == EVAL_DIAG)
        ILQR_LAUNCH_FW(DiagQR, false);
    else if (ev == EVAL_DENSE && st)
        ILQR_LAUNCH_FW(DenseQR, true);
    else if (ev == EVAL_DENSE)
        ILQR_LAUNCH_FW(DenseQR, false);
    else if (st)
        ILQR_LAUNCH_FW(PathQR, true);
    else
        ILQR_LAUNCH_FW(PathQR, false);
#undef ILQR_LAUNCH_FW
    return hip_status(hipGetLastError());
}

// rows of the per-problem goals in the solve's workspace (0: shared goals)
static int64_t goal_rows_x(const ilqr_model* m)
{
    return !per_problem_goals(m) ? 0 : (m->cost_kind == ILQR_COST_PATH_QR ? ((int64_t)m->T + 1) * 5 : 5);
}
static int64_t goal_rows_u(const ilqr_model* m)
{
    return !per_problem_goals(m) ? 0 : (m->cost_kind == ILQR_COST_PATH_QR ? (int64_t)m->T : 1);
}

size_t ilqr_solve_workspace_bytes(const ilqr_model* model, int64_t B)
{
    if (check_model(model) != ILQR_OK || B < 1)
        return 0;
    int64_t ld = ilqr_bm_ld(B), T = model->T;
    int64_t rows = 2 * ((T + 1) * 5 + T + T + T * 5) + goal_rows_x(model) + goal_rows_u(model);
    return (size_t)rows * ld * sizeof(double) + 2 * (size_t)ld * sizeof(int32_t);
}

// The solve on problems idx[0..B-1] of the caller's per-problem ARRAYS (x0, us_init, k, K, xs, us: row idx[b]
// of each belongs to problem b of this launch; idx == nullptr: row b), with the per-problem SCALARS (delta, J,
// mu, iters, status, trace) compact, [B].  The chunked solve continues its running problems this way without
// staging copies of their trajectories and gains.
// Compute units of the current device, looked up once per device and process (the launch-shape rules below are
// on the path of every solve, closed-loop re-solves of ~0.1 ms included).
static int device_cu_count()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return 256;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v > 0)
        return v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
        return 256;
    cached[dev].store(v, std::memory_order_relaxed);
    return v;
}

static int solve_batched_indexed(const ilqr_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                                 const double* us_init, double* delta, double* k, double* K, double* xs,
                                 double* us, double* J, double* mu, int32_t* iters, int32_t* status,
                                 const ilqr_trace* trace, void* workspace, size_t workspace_bytes, void* stream,
                                 const int64_t* idx)
{
    int rc = check_model(model);
    if (rc != ILQR_OK)
        return rc;
    if (!x0 || !us_init || !delta || !k || !K || !xs || !us || !J || !mu || !iters || !status ||
        !workspace || B < 1)
        return ILQR_ERR_ARG;
    if (workspace_bytes < ilqr_solve_workspace_bytes(model, B))
        return ILQR_ERR_WORKSPACE;
    CartpoleP mp;
    QRCostP C;
    BoundsP bd;
    SolveOptsP O;
    fill_model(model, mp, C, bd);
    rc = fill_opts(opts, O);
    if (rc != ILQR_OK)
        return rc;
    hipStream_t st = as_stream(stream);
    int64_t ld = ilqr_bm_ld(B), T = model->T;
    SolveBuffers W;
    double* p = static_cast<double*>(workspace);
    for (int c = 0; c < 2; ++c) {
        W.xs[c] = p; p += (T + 1) * 5 * ld;
        W.us[c] = p; p += T * ld;
        W.k[c] = p;  p += T * ld;
        W.K[c] = p;  p += T * 5 * ld;
    }
    W.goal_rows_x = goal_rows_x(model);
    W.goal_rows_u = goal_rows_u(model);
    double* const goal_x = p; p += W.goal_rows_x * ld;
    double* const goal_u = p; p += W.goal_rows_u * ld;
    W.goal_x = per_problem_goals(model) ? goal_x : nullptr;
    W.goal_u = per_problem_goals(model) ? goal_u : nullptr;
    W.sel_traj = reinterpret_cast<int32_t*>(p);
    W.sel_gain = W.sel_traj + ld;

    // into the wave-tiled workspace, one launch: x0 goes to rows 0..4 of the (T+1)*5-row state array of each tile;
    // goals that differ from problem to problem travel the same way
    {
        PackJobs Jb;
        const int nj = per_problem_goals(model) ? 6 : 4;
        const double* srcs[PACK_JOBS] = {x0, us_init, k, K, model->x_goal, model->u_goal};
        double* dsts[PACK_JOBS] = {W.xs[0], W.us[0], W.k[0], W.K[0], goal_x, goal_u};
        const int64_t rows[PACK_JOBS] = {5, T, T, T * 5, W.goal_rows_x, W.goal_rows_u};
        const int64_t strides[PACK_JOBS] = {5, T, T, T * 5, model->goal_stride_x, model->goal_stride_u};
        const int64_t rows_dst[PACK_JOBS] = {(T + 1) * 5, T, T, T * 5, W.goal_rows_x, W.goal_rows_u};
        unsigned y = 0;
        for (int i = 0; i < PACK_JOBS; ++i) {
            const bool on = i < nj;
            Jb.src[i] = on ? srcs[i] : nullptr;
            Jb.dst[i] = on ? dsts[i] : nullptr;
            Jb.R[i] = on ? rows[i] : 0;
            Jb.S[i] = on ? strides[i] : 0;
            Jb.Rd[i] = on ? rows_dst[i] : 0;
            Jb.y0[i] = y;
            if (on)
                y += (unsigned)((rows[i] + 63) / 64);
        }
        Jb.y0[PACK_JOBS] = y;
        Jb.n = nj;
        Jb.idx = idx;
        hipLaunchKernelGGL(k_pack_jobs, dim3((unsigned)((B + 63) / 64), y), dim3(256), 0, st, Jb, B);
    }
    ilqr_trace tr;
    if (trace)
        tr = *trace;
    else
        std::memset(&tr, 0, sizeof(tr));
    // Launch shape.  Measured with per-wave timestamps at 16384 problems: when every CU runs one wave,
    // waves on certain CUs (the same hardware positions in every run) take up to 10 % longer than the
    // median and set the kernel time (72 ms); packed four to a CU on a quarter of the CUs, the same waves
    // finish within 3 % of each other (66.6 ms).  So long solves use 4-wave workgroups at every batch
    // size, and 32 problems per wavefront while every wave still finds a free SIMD (1024 of them): a
    // wave-level slow path is entered when any lane needs it, and an instruction costs the same for 32
    // lanes as for 64 (65.9 ms; 16 per wave is no faster and far slower once waves outnumber the SIMDs).
    // Short launches (the re-solves of a closed loop: T = 15, 10 iterations) measured 8 % faster with one
    // full wave per workgroup and keep that shape.
    const bool short_launch = (int64_t)T * O.max_iter < 2048;
    int lpw = (B <= 32 * 1024 && !short_launch) ? 32 : WAVE;
    // Solves that leave SIMDs idle share each group's backward sweep and first rollout between several wavefronts
    // of one workgroup (see pair_master_sweep).  A workgroup's wavefronts need a SIMD each on ONE CU (the kernel's
    // register count allows one wavefront per SIMD), so a CU holds one group of three (master, helper, verifier)
    // or two groups of two; groups that find no CU would wait for a second round.  Hence: three wavefronts per
    // group while the groups number at most the CUs, two while at most twice the CUs; 32 problems per group while
    // even that many groups fit, 64 otherwise.  ILQR_SOLVE_GROUP=1/2/3 overrides the choice.
    const int n_cu = device_cu_count();
    const int64_t groups64 = (B + 63) / 64, groups32 = (B + 31) / 32;
    int group = 1; // wavefronts per group of problems
    if (groups64 <= n_cu)
        group = 3;
    else if (groups64 <= 2 * (int64_t)n_cu)
        group = 2;
    if (const char* e = std::getenv("ILQR_SOLVE_GROUP")) {
        int v = std::atoi(e);
        group = (v >= 1 && v <= 3) ? v : group;
    }
    const bool pair = group >= 2;
    if (pair)
        lpw = groups32 <= (group == 3 ? 1 : 2) * (int64_t)n_cu ? 32 : WAVE;
    if (const char* e = std::getenv("ILQR_SOLVE_LPW")) {
        int v = std::atoi(e);
        lpw = (v == 16 || v == 32) ? v : WAVE;
    }
    O.lanes_per_wave = lpw;
    const int64_t nwaves = (B + lpw - 1) / lpw;
    // Waves that share a workspace tile (64 / lpw of them) must sit in one workgroup, i.e. on one CU:
    // sharing a tile across XCDs was measured 10 % slower.
    int blk = (lpw == WAVE && short_launch && nwaves <= 256) ? WAVE : 256;
    if (const char* e = std::getenv("ILQR_SOLVE_BLOCK")) {
        int v = std::atoi(e);
        if (v == 256 || v == 128 || (v == WAVE && lpw == WAVE))
            blk = v;
    }
    dim3 grid((unsigned)((nwaves * WAVE + blk - 1) / blk));
    if (pair) {
        blk = group * WAVE;
        grid = dim3((unsigned)((nwaves + 15) / 16 * 16)); // whole chunks of 16 groups (the kernel's group <-> workgroup map)
    }
    int ev = cost_eval(C);
#define ILQR_LAUNCH_SOLVE(COST)                                                                                  \
    do {                                                                                                         \
        if (group == 3)                                                                                          \
            hipLaunchKernelGGL((k_solve<COST, 3>), grid, dim3(blk), 0, st, mp, C, O, bd, (int)T, B, W, delta, J, mu,    \
                               iters, status, tr);                                                               \
        else if (group == 2)                                                                                     \
            hipLaunchKernelGGL((k_solve<COST, 2>), grid, dim3(blk), 0, st, mp, C, O, bd, (int)T, B, W, delta, J, mu,    \
                               iters, status, tr);                                                               \
        else                                                                                                     \
            hipLaunchKernelGGL((k_solve<COST, 1>), grid, dim3(blk), 0, st, mp, C, O, bd, (int)T, B, W, delta, J,        \
                               mu, iters, status, tr);                                                           \
    } while (0)
    if (ev == EVAL_DIAG)
        ILQR_LAUNCH_SOLVE(DiagQR);
    else if (ev == EVAL_DENSE)
        ILQR_LAUNCH_SOLVE(DenseQR);
    else
        ILQR_LAUNCH_SOLVE(PathQR);
#undef ILQR_LAUNCH_SOLVE
    {
        UnpackJobs Jb;
        const double* sa[4] = {W.xs[0], W.us[0], W.k[0], W.K[0]};
        const double* sb[4] = {W.xs[1], W.us[1], W.k[1], W.K[1]};
        const int32_t* sl[4] = {W.sel_traj, W.sel_traj, W.sel_gain, W.sel_gain};
        double* dsts[4] = {xs, us, k, K};
        const int64_t rows[4] = {(T + 1) * 5, T, T, T * 5};
        unsigned y = 0;
        for (int i = 0; i < 4; ++i) {
            Jb.srcA[i] = sa[i];
            Jb.srcB[i] = sb[i];
            Jb.sel[i] = sl[i];
            Jb.dst[i] = dsts[i];
            Jb.R[i] = rows[i];
            Jb.y0[i] = y;
            y += (unsigned)((rows[i] + 63) / 64);
        }
        Jb.y0[4] = y;
        Jb.idx = idx;
        hipLaunchKernelGGL(k_unpack_jobs, dim3((unsigned)((B + 63) / 64), y), dim3(256), 0, st, Jb, B);
    }
    return hip_status(hipGetLastError());
}

int ilqr_solve_batched(const ilqr_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                       const double* us_init, double* delta, double* k, double* K, double* xs,
                       double* us, double* J, double* mu, int32_t* iters, int32_t* status,
                       const ilqr_trace* trace, void* workspace, size_t workspace_bytes, void* stream)
{
    return solve_batched_indexed(model, opts, B, x0, us_init, delta, k, K, xs, us, J, mu, iters, status, trace, workspace,
                                 workspace_bytes, stream, nullptr);
}

} // extern "C" (interrupted for the library-owned workspace of ilqr_solve_batched_static)

// The workspace of ilqr_solve_batched_static: device memory that belongs to the library, one copy per GPU the
// library is loaded on, held until the process ends.  In host code the array's NAME is only a handle for
// hipGetSymbolAddress; the address the kernels are given is looked up once per device and kept in
// g_static_ws[device].
static __device__ __align__(256) unsigned char g_static_scratch[ILQR_STATIC_WORKSPACE_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
void* g_static_ws[MAX_DEVICES]; // device address of g_static_scratch on each device, null until looked up
std::mutex g_static_ws_mutex;

// Address of the current device's copy of g_static_scratch (looked up on the first call on that device).
int static_workspace(void** out)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES)
        return ILQR_ERR_HIP;
    std::lock_guard<std::mutex> lock(g_static_ws_mutex);
    if (g_static_ws[dev] == nullptr) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch)) != hipSuccess || p == nullptr)
            return ILQR_ERR_HIP;
        g_static_ws[dev] = p;
    }
    *out = g_static_ws[dev];
    return ILQR_OK;
}
} // namespace

extern "C" {

int ilqr_solve_batched_static(const ilqr_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                              const double* us_init, double* delta, double* k, double* K, double* xs, double* us,
                              double* J, double* mu, int32_t* iters, int32_t* status, const ilqr_trace* trace,
                              void* stream)
{
    int rc = check_model(model);
    if (rc != ILQR_OK)
        return rc;
    if (B < 1)
        return ILQR_ERR_ARG;
    // every array of the launch lives in the fixed workspace: larger problems are refused, not truncated
    if (ilqr_solve_workspace_bytes(model, B) > (size_t)ILQR_STATIC_WORKSPACE_BYTES)
        return ILQR_ERR_WORKSPACE;
    void* ws = nullptr;
    rc = static_workspace(&ws);
    if (rc != ILQR_OK)
        return rc;
    // Nothing in the workspace has to start at zero: the conversion kernel writes x0, the controls and the
    // gains of the B problems of this call before the solve kernel reads them, the second copies and the
    // selectors are written before they are read, and columns beyond B are never read, so what an earlier
    // (larger) call has left behind is dead data.
    return solve_batched_indexed(model, opts, B, x0, us_init, delta, k, K, xs, us, J, mu, iters, status, trace, ws,
                                 (size_t)ILQR_STATIC_WORKSPACE_BYTES, stream, nullptr);
}


} // extern "C" (interrupted for the helper kernels of the chunked solve)

namespace {

// dst[j][:] = src[idx[j]][:] and back, rows of `row` doubles (reference-layout per-problem arrays)
__global__ __launch_bounds__(256) void k_gather_rows(const double* __restrict__ src, double* __restrict__ dst,
                                                     const int64_t* __restrict__ idx, int64_t n, int64_t row)
{
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * row)
        return;
    int64_t j = e / row, c = e - j * row;
    dst[e] = src[idx[j] * row + c];
}
__global__ __launch_bounds__(256) void k_scatter_rows(const double* __restrict__ src, double* __restrict__ dst,
                                                      const int64_t* __restrict__ idx, int64_t n, int64_t row)
{
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * row)
        return;
    int64_t j = e / row, c = e - j * row;
    dst[idx[j] * row + c] = src[e];
}
// iteration counts add up over the launches, the status is that of the last launch a problem took part in
__global__ __launch_bounds__(256) void k_chunk_bookkeeping(const int32_t* __restrict__ iters_c,
                                                           const int32_t* __restrict__ status_c,
                                                           const int64_t* __restrict__ idx, int64_t n,
                                                           int32_t* __restrict__ iters, int32_t* __restrict__ status)
{
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n)
        return;
    int64_t b = idx ? idx[j] : j;
    iters[b] += iters_c[j];
    status[b] = status_c[j];
}
__global__ __launch_bounds__(256) void k_chunk_init(int64_t B, double* __restrict__ mu, int32_t* __restrict__ iters,
                                                    int32_t* __restrict__ status)
{
    int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B)
        return;
    mu[b] = 1.0; // update_u resets mu per call (ilqr_core.py:264)
    iters[b] = 0;
    status[b] = ILQR_RAN_OUT;
}

inline unsigned blocks256(int64_t n) { return (unsigned)((n + 255) / 256); }

struct ChunkBuffers { // carved out of the caller's workspace behind the solve workspace
    double *delta, *mu, *J; // compact per-problem scalars of the running problems
    double *dJ, *dJ_all;    // Delta_J1 / Delta_J2 of the last backward pass: [2, n] of the running problems, [2, B] of all
    int32_t *iters, *status;
    int64_t* idx;
};

} // namespace

extern "C" {

size_t ilqr_solve_chunked_workspace_bytes(const ilqr_model* model, int64_t B)
{
    size_t base = ilqr_solve_workspace_bytes(model, B);
    if (base == 0)
        return 0;
    // compact copies of the per-problem scalars (delta, mu, J, Delta_J1, Delta_J2), Delta_J1 / Delta_J2 of every
    // problem, iteration counts, status and the index list
    size_t doubles = (size_t)B * 7;
    return base + doubles * sizeof(double) + (size_t)B * (2 * sizeof(int32_t) + sizeof(int64_t)) + 256;
}

int ilqr_solve_batched_chunked(const ilqr_model* model, const ilqr_opts* opts, int32_t chunk, int64_t B,
                               const double* x0, const double* us_init, double* delta, double* k, double* K,
                               double* xs, double* us, double* J, double* mu, int32_t* iters, int32_t* status,
                               void* workspace, size_t workspace_bytes, void* stream)
{
    int rc = check_model(model);
    if (rc != ILQR_OK)
        return rc;
    if (!opts || chunk < 1 || !x0 || !us_init || !delta || !k || !K || !xs || !us || !J || !mu || !iters ||
        !status || !workspace || B < 1)
        return ILQR_ERR_ARG;
    if (workspace_bytes < ilqr_solve_chunked_workspace_bytes(model, B))
        return ILQR_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    const int64_t T = model->T;
    const size_t base = (ilqr_solve_workspace_bytes(model, B) + 255) / 256 * 256;
    ChunkBuffers G;
    {
        double* p = reinterpret_cast<double*>(static_cast<char*>(workspace) + base);
        G.delta = p; p += B;
        G.mu = p; p += B;
        G.J = p; p += B;
        G.dJ = p; p += 2 * B;
        G.dJ_all = p; p += 2 * B;
        G.idx = reinterpret_cast<int64_t*>(p);
        G.iters = reinterpret_cast<int32_t*>(G.idx + B);
        G.status = G.iters + B;
    }
    ilqr_opts o = *opts;
    o.mu_init = 0.0; // every launch starts from the mu array ...
    // ... and from the Delta_J sums of the problem's last backward pass (zeros at the start: ilqr_core.py:278-279)
    if (hipMemsetAsync(G.dJ_all, 0, (size_t)2 * B * sizeof(double), st) != hipSuccess)
        return ILQR_ERR_HIP;
    hipLaunchKernelGGL(k_chunk_init, dim3(blocks256(B)), dim3(256), 0, st, B, mu, iters, status);
    std::vector<int32_t> h_status((size_t)B);
    std::vector<int64_t> h_idx;
    int64_t n = B;      // problems still running
    bool full = true;   // all of them: the launches work on the caller's arrays in place
    int done = 0;
    const int max_iter = opts->max_iter;
    // Problems one round of wavefronts holds (one wavefront of 64 problems per SIMD).  A launch with slightly
    // more runs a second round for the overhang that costs most of a full one (67185 problems take as long
    // as 77654), so while the running set is within a sixth of a round above a whole number of rounds the
    // chunks are a quarter as long: the set is re-counted sooner and drops a round earlier.
    const int64_t round_problems = (int64_t)device_cu_count() * 4 * WAVE;
    for (;;) {
        const int64_t over = n % round_problems; // problems beyond the last full round
        const bool overhang = n > round_problems && over > 0 && over <= round_problems / 6;
        o.max_iter = std::min<int>(overhang ? std::max(1, chunk / 4) : chunk, max_iter - done);
        if (full) {
            // (us may alias us_init from the second launch on: the input is packed before the output is written)
            o.dJ_carry = G.dJ_all;
            rc = ilqr_solve_batched(model, &o, B, x0, done == 0 ? us_init : us, delta, k, K, xs, us, J, mu, G.iters,
                                    G.status, nullptr, workspace, base, stream);
            if (rc != ILQR_OK)
                return rc;
            hipLaunchKernelGGL(k_chunk_bookkeeping, dim3(blocks256(B)), dim3(256), 0, st, G.iters, G.status,
                               (const int64_t*)nullptr, B, iters, status);
        } else {
            auto gather = [&](const double* src, double* dst, int64_t row) {
                hipLaunchKernelGGL(k_gather_rows, dim3(blocks256(n * row)), dim3(256), 0, st, src, dst, G.idx, n, row);
            };
            auto scatter = [&](const double* src, double* dst, int64_t row) {
                hipLaunchKernelGGL(k_scatter_rows, dim3(blocks256(n * row)), dim3(256), 0, st, src, dst, G.idx, n, row);
            };
            // trajectories and gains stay where they are (the conversion kernels of the solve follow the index
            // list); only the per-problem scalars are compacted
            gather(delta, G.delta, 1);
            gather(mu, G.mu, 1);
            gather(G.dJ_all, G.dJ, 1);
            gather(G.dJ_all + B, G.dJ + n, 1);
            o.dJ_carry = G.dJ; // [2, n]
            rc = solve_batched_indexed(model, &o, n, x0, us, G.delta, k, K, xs, us, G.J, G.mu, G.iters, G.status, nullptr,
                                       workspace, base, stream, G.idx);
            if (rc != ILQR_OK)
                return rc;
            scatter(G.J, J, 1);
            scatter(G.delta, delta, 1);
            scatter(G.mu, mu, 1);
            scatter(G.dJ, G.dJ_all, 1);
            scatter(G.dJ + n, G.dJ_all + B, 1);
            hipLaunchKernelGGL(k_chunk_bookkeeping, dim3(blocks256(n)), dim3(256), 0, st, G.iters, G.status,
                               (const int64_t*)G.idx, n, iters, status);
        }
        done += o.max_iter;
        if (done >= max_iter)
            break;
        // who is still running?  (one small read-back per chunk: the next launch is sized on the host)
        if (hipMemcpyAsync(h_status.data(), G.status, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st) !=
                hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return ILQR_ERR_HIP;
        std::vector<int64_t> next;
        next.reserve((size_t)n);
        for (int64_t j = 0; j < n; ++j)
            if (h_status[(size_t)j] == ILQR_RAN_OUT)
                next.push_back(full ? j : h_idx[(size_t)j]);
        if (full && (int64_t)next.size() == B)
            continue;
        full = false;
        h_idx.swap(next);
        n = (int64_t)h_idx.size();
        if (n == 0)
            break;
        if (hipMemcpyAsync(G.idx, h_idx.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) // (h_idx is reused by the next round)
            return ILQR_ERR_HIP;
    }
    return hip_status(hipGetLastError());
}

int ilqr_boxqp_batched(int64_t B, int32_t m, const double* H, const double* g, const double* lower,
                       const double* upper, const double* x0, const ilqr_opts* opts, double* x,
                       int32_t* result, double* Hff, int32_t* hff_dim, int32_t* free_mask,
                       int32_t* counters, void* stream)
{
    if (!H || !g || !lower || !upper || !x || !result || !Hff || !hff_dim || !free_mask || !counters || B < 1)
        return ILQR_ERR_ARG;
    if (m < 1 || m > QM)
        return ILQR_ERR_UNSUPPORTED;
    QPOptsP q;
    int rc = fill_qp(opts, q);
    if (rc != ILQR_OK)
        return rc;
    if (m == 1)
        hipLaunchKernelGGL(k_boxqp1, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, as_stream(stream), q, B, H,
                           g, lower, upper, x0, x, result, Hff, hff_dim, free_mask, counters);
    else
        hipLaunchKernelGGL(k_boxqp, dim3((unsigned)((B + WAVE - 1) / WAVE)), dim3(WAVE), 0, as_stream(stream), q,
                           B, (int)m, H, g, lower, upper, x0, x, result, Hff, hff_dim, free_mask, counters);
    return hip_status(hipGetLastError());
}

#ifdef ILQR_PAIR_PROF
int ilqr_debug_pair_prof(unsigned long long* host32, int reset)
{
    unsigned long long z[32] = {0};
    if (reset)
        return hip_status(hipMemcpyToSymbol(HIP_SYMBOL(g_pair_prof), z, sizeof(z)));
    return hip_status(hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_pair_prof), sizeof(z)));
}
#endif

#ifdef ILQR_QP_PROF
int ilqr_debug_qp_wave(unsigned int* host2x8192, int reset)
{
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(ilqr::g_qp_wave)) != hipSuccess)
            return ILQR_ERR_HIP;
        return hip_status(hipMemset(p, 0, sizeof(unsigned int) * 2 * 8192));
    }
    return hip_status(hipMemcpyFromSymbol(host2x8192, HIP_SYMBOL(ilqr::g_qp_wave), sizeof(unsigned int) * 2 * 8192));
}

int ilqr_debug_qp_prof(unsigned long long* host16, int reset)
{
    unsigned long long z[16] = {0};
    if (reset)
        return hip_status(hipMemcpyToSymbol(HIP_SYMBOL(ilqr::g_qp_prof), z, sizeof(z)));
    return hip_status(hipMemcpyFromSymbol(host16, HIP_SYMBOL(ilqr::g_qp_prof), sizeof(z)));
}
#endif

int ilqr_debug_wave_log(void* device_buffer)
{
    unsigned long long* p = static_cast<unsigned long long*>(device_buffer);
    return hip_status(hipMemcpyToSymbol(HIP_SYMBOL(g_wave_log), &p, sizeof(p)));
}

int ilqr_cartpole_plant_step(int64_t B, const double* par, const double* st, const double* u, double* out,
                             void* stream)
{
    if (!par || !st || !u || !out || B < 1)
        return ILQR_ERR_ARG;
    PlantP p{par[0], par[1], par[2], par[3], par[4]};
    hipLaunchKernelGGL(k_plant_step, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, as_stream(stream), p,
                       B, st, u, out);
    return hip_status(hipGetLastError());
}

int ilqr_cartpole_mpc_advance(int64_t B, int64_t N, const double* par, double* st, const double* us_solved,
                              double* us_seed, double* u_applied, double* x0_aug, void* stream)
{
    if (!par || !st || !us_solved || !us_seed || !u_applied || !x0_aug || B < 1 || N < 1 || us_seed == us_solved)
        return ILQR_ERR_ARG;
    PlantP p{par[0], par[1], par[2], par[3], par[4]};
    hipLaunchKernelGGL(k_mpc_advance, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, as_stream(stream), p, B, N,
                       st, us_solved, us_seed, u_applied, x0_aug);
    return hip_status(hipGetLastError());
}

int ilqr_detmath(int64_t n, const double* a, const double* b, double* s, double* c, double* t, double* sq,
                 double* dv, void* stream)
{
    if (!a || !b || !s || !c || !t || !sq || !dv || n < 1)
        return ILQR_ERR_ARG;
    hipLaunchKernelGGL(k_detmath, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), n, a,
                       b, s, c, t, sq, dv);
    return hip_status(hipGetLastError());
}

} // extern "C"
