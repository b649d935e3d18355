/*
 * C ABI of the MI355X batched iLQR library (libilqr_hip.so).
 *
 * The reference (ZhengYi0310/ilqr) is pure Python: its "FFI" for this path is
 * the set of Python methods cited beside each entry point below.  Every entry
 * point takes plain pointers and sizes only.  Unless stated otherwise pointers
 * are DEVICE pointers to fp64 / int32 arrays owned by the caller, the call is
 * asynchronous on `stream` (a hipStream_t passed as void*, NULL = default
 * stream), re-entrant (all state lives in caller buffers) and returns ILQR_OK
 * or a negative ilqr_status; per-problem numerical outcomes are written to
 * output arrays, never to the return code.
 *
 * Layouts.  "ref" layout is the reference's own, batch outermost:
 *   x0 [B,n]  us [B,T,m]  xs [B,T+1,n]  k [B,T,m]  K [B,T,m,n].
 * "bm" (batch-minor) layout is what the kernels stream: the per-problem array
 * is flattened row-major to R numbers and stored as [R, ld] with the problem
 * index innermost, ld >= B (ilqr_bm_ld(B) rounds B up to a multiple of 64) so
 * that the 64 lanes of a wavefront read 64 consecutive doubles.
 */
#ifndef ILQR_HIP_H
#define ILQR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ilqr_status {
    ILQR_OK = 0,
    ILQR_ERR_ARG = -1,         /* null pointer, non-positive size, ld < B ... */
    ILQR_ERR_UNSUPPORTED = -2, /* model / dimension / option without a kernel */
    ILQR_ERR_HIP = -3,         /* a HIP runtime call failed */
    ILQR_ERR_WORKSPACE = -4    /* workspace too small */
} ilqr_status;

enum { ILQR_MODEL_CARTPOLE = 0 };
enum { ILQR_COST_QR = 0, ILQR_COST_PATH_QR = 1 };

/* Per-problem status written by ilqr_solve_batched (update_u's three exits,
 * ilqr_core.py:284, :331/:347, :337-339). */
enum { ILQR_RAN_OUT = 0, ILQR_CONVERGED = 1, ILQR_MU_MAX = 2 };

/* Model + cost + bounds shared by every problem of a batch.  All pointers in
 * this struct are HOST pointers; the (small) arrays are copied into kernel
 * arguments by each call.
 *   cartpole: CartPoleDynamics(dt, max_control, mc, mp, l, g), ilqr/cartpole.py:8;
 *             n = 5 ([x, x', sin, cos, theta']), m = 1.
 *   QRCost(Q, R, Q_terminal, x_goal, u_goal), ilqr/ilqr_cost.py:490-530.
 *   PathQRCost(Q, R, x_path, u_path, Q_terminal), ilqr/ilqr_cost.py:639-680
 *   (cost_kind = ILQR_COST_PATH_QR): time-indexed goals shared by the batch; the
 *   one exception to the host-pointer rule: x_goal / u_goal then point to DEVICE
 *   arrays x_path [(T+1)*n] and u_path [T*m] (both required), read by the kernels. */
typedef struct ilqr_model {
    int32_t model;         /* ILQR_MODEL_* */
    int32_t n, m, T;       /* state dim, control dim, horizon N */
    int32_t cost_kind;     /* ILQR_COST_* */
    int32_t control_limit; /* iLQR(control_limit=...), ilqr_core.py:32; only 1 is supported */
    double dt, mc, mp, l, g;
    const double *Q, *R, *Q_terminal; /* [n*n], [m*m], [n*n] row-major */
    const double *x_goal, *u_goal;    /* QR: host [n], [m] (NULL = zeros); PATH_QR: device [(T+1)*n], [T*m] */
    const double *lower, *upper;      /* dynamics.lower_bound / upper_bound, [m] */
    /* Per-problem goals.  In the reference every problem is its own solver object with its own cost
     * (ilqr_cost.py:490-530, :639-770), so B independent calls may each track their own goal or path.
     * goal_stride_x = goal_stride_u = 0: the goals above, shared by the batch.  Non-zero (both): x_goal and
     * u_goal are DEVICE arrays holding one goal (QR: n resp. m numbers) or one path (PATH_QR: (T+1)*n resp. T*m
     * numbers) per problem, problem b's at x_goal + b * goal_stride_x and u_goal + b * goal_stride_u (strides
     * in doubles, at least the size of one goal / path).  The solve entry points accept them (the single-launch,
     * chunked and static ones); the stage entry points (*_bm) return ILQR_ERR_UNSUPPORTED.
     * (ilqr_version() >= 102; zero-initialise the struct so that callers built before keep shared goals.) */
    int64_t goal_stride_x, goal_stride_u;
} ilqr_model;

/* update_u keyword arguments (ilqr_core.py:241), the solver constants of
 * iLQR.__init__ (:58-62) and the boxQP options the solver passes (:154-155). */
typedef struct ilqr_opts {
    int32_t max_iter; /* n_interations */
    int32_t n_alpha;  /* line_search_steps, <= 32 */
    const double* alphas; /* HOST [n_alpha]: 1.1 ** (-arange(n_alpha) ** 2), :267 */
    double tol, scale;
    double mu_min, mu_max, delta0;
    int32_t qp_max_iter;
    int32_t pad_;
    double qp_min_grad, qp_min_rel_improve, qp_step_dec, qp_min_step, qp_armijo;
    /* update_u resets the regularisation to 1.0 on entry (:264): pass a negative
     * value (or NaN) for that.  A value >= 0 starts every problem from mu_init[b]
     * read from the `mu` output array instead (in/out), which lets a caller run
     * the loop one iteration per call (on_iteration callbacks, :342-345). */
    double mu_init;
    /* DEVICE [2, B] or NULL.  Delta_J1 and Delta_J2 of update_u (ilqr_core.py:278-279, :289) are loop-carried:
     * after a backward pass that gives up at mu_max (:291-299) the line search runs on the sums of the previous
     * iteration (:305).  A caller that cuts one solve into several calls (mu_init >= 0) hands this array to each
     * of them: it is read on entry and written on exit (Delta_J1 of problem b at [b], Delta_J2 at [B + b]) ONLY
     * when mu_init >= 0; a fresh solve (mu_init < 0) starts from zeros (:278-279) and never touches the array.
     * The struct has grown by this trailing field (ilqr_version() >= 101): zero-initialise every ilqr_opts
     * (`ilqr_opts o = {0};` or memset) before filling it, so that fields a caller does not know stay NULL / 0. */
    double* dJ_carry;
} ilqr_opts;

/* Optional per-iteration record of ilqr_solve_batched, each array [max_iter, B]
 * (iteration outermost), any pointer may be NULL.  It carries what update_u's
 * on_iteration callback sees (ilqr_core.py:342-345) plus the counters needed
 * to compare line-search and boxQP behaviour. */
typedef struct ilqr_trace {
    double* J;             /* J_opt after the iteration */
    int32_t* accepted;     /* 0/1; -1 where the problem was already finished */
    int32_t* alpha_idx;    /* index of the alpha at which the line search ended (n_alpha = ran out) */
    double* mu;
    double* delta;
    int32_t* qp_iter;      /* boxQP search steps summed over the T steps of the backward pass */
    int32_t* qp_backtrack; /* Armijo step reductions summed over the backward pass */
    int32_t* n_clamped;    /* steps of the backward pass with a clamped control */
    int32_t* stopped_at;   /* step at which Q_uu was not PD (ilqr_core.py:137-140), -1 if none */
} ilqr_trace;

int ilqr_version(void);
const char* ilqr_status_string(int status);

/* Leading dimension the library recommends for B problems (B rounded up to 64). */
int64_t ilqr_bm_ld(int64_t B);

/* ref [B, R] -> bm [R, ld] and back.  No counterpart in the reference. */
int ilqr_pack_bm(const double* src, double* dst_bm, int64_t B, int64_t R, int64_t ld, void* stream);
int ilqr_unpack_bm(const double* src_bm, double* dst, int64_t B, int64_t R, int64_t ld, void* stream);

/* iLQR._dynamics_rollout (ilqr_core.py:69-83) + iLQR._trajectory_cost (:230-239).
 * x0_bm [n, ld], us_bm [T*m, ld] -> xs_bm [(T+1)*n, ld], J [B]. */
int ilqr_rollout_cost_bm(const ilqr_model* model, int64_t B, int64_t ld, const double* x0_bm,
                         const double* us_bm, double* xs_bm, double* J, void* stream);

/* Dynamics.f_x / f_u (ilqr_dynamics.py:498-524; Theano Jacobian of diff_utils.py:28-37)
 * and Cost.l_x, l_u, l_xx, l_uu, l_ux (ilqr_cost.py:555-632) for every (problem, step),
 * plus the terminal l_x, l_xx: the tensors _compute_Q_terms (ilqr_core.py:189-198) asks
 * for, written as batch-minor streams:
 *   fx [T*n*n, ld]  fu [T*n*m, ld]  lx [T*n, ld]  lu [T*m, ld]  lxx [T*n*n, ld]
 *   luu [T*m*m, ld]  lux [T*m*n, ld]  lxT [n, ld]  lxxT [n*n, ld]. */
typedef struct ilqr_deriv_bm {
    double *fx, *fu, *lx, *lu, *lxx, *luu, *lux, *lxT, *lxxT;
} ilqr_deriv_bm_t;
int ilqr_deriv_bm(const ilqr_model* model, int64_t B, int64_t ld, const double* xs_bm,
                  const double* us_bm, const ilqr_deriv_bm_t* out, void* stream);

/* iLQR._backward_pass + _compute_Q_terms (ilqr_core.py:109-228) with boxQP inside
 * (ilqr_boxQP.py:9-186), one problem per lane.
 *   mu [B]; deriv: NULL = evaluate derivatives in the kernel (the fused sweep the solver
 *   uses), else read the streams written by ilqr_deriv_bm (the staged sweep; xs_bm may
 *   then be NULL).
 *   -> k_bm [T*m, ld], K_bm [T*m*n, ld] (zero where the sweep did not reach),
 *      dJ [2, ld] (Delta_J1, Delta_J2),
 *      info [5, ld] int32: stopped_at, qp_iter, qp_backtrack, n_clamped, rc (0 ok, 1 not PD, 2 non-finite Q_uu),
 *      qp_result_bm [T, ld] (boxQP result code per step, -100 = not reached; may be NULL),
 *      free_bm [T*m, ld] (boxQP free mask per step; may be NULL). */
int ilqr_backward_bm(const ilqr_model* model, const ilqr_opts* opts, int64_t B, int64_t ld,
                     const double* xs_bm, const double* us_bm, const double* mu,
                     const ilqr_deriv_bm_t* deriv, double* k_bm, double* K_bm, double* dJ,
                     int32_t* info, int32_t* qp_result_bm, int32_t* free_bm, void* stream);

/* iLQR._forward_pass (ilqr_core.py:85-107) + _trajectory_cost for every
 * (problem, alpha) in parallel: alphas HOST [n_alpha] -> J_new [n_alpha, ld];
 * xs_new_bm [n_alpha, (T+1)*n, ld] and us_new_bm [n_alpha, T*m, ld] are written
 * when not NULL. */
int ilqr_forward_ls_bm(const ilqr_model* model, int64_t B, int64_t ld, int32_t n_alpha,
                       const double* alphas, const double* xs_bm, const double* us_bm,
                       const double* k_bm, const double* K_bm, double* J_new, double* xs_new_bm,
                       double* us_new_bm, void* stream);

/* boxQP(H, g, lower, upper, x0, **opts) (ilqr_boxQP.py:9-186) for B independent QPs,
 * ref layout: H [B,m,m], g/lower/upper/x0 [B,m] (x0 may be NULL: start at the box
 * centre, :44-46), m <= 8.  Only the qp_* fields of opts are read.
 *   -> x [B,m], result [B] (codes of :173-180), Hff [B,m,m] (free-subspace Cholesky
 *      factor in the leading hff_dim x hff_dim block, row stride m; all ones if no
 *      factorisation happened, :36), hff_dim [B], free_mask [B,m],
 *      counters [B,3] = search steps, factorisations, Armijo reductions.
 *   result = -99 marks a QP on which the reference would raise LinAlgError
 *   (non-finite H, or a free sub-block that is not positive definite). */
int ilqr_boxqp_batched(int64_t B, int32_t m, const double* H, const double* g, const double* lower,
                       const double* upper, const double* x0, const ilqr_opts* opts, double* x,
                       int32_t* result, double* Hff, int32_t* hff_dim, int32_t* free_mask,
                       int32_t* counters, void* stream);

/* iLQR.update_u (ilqr_core.py:241-399) for B problems, entirely on the device: one
 * launch runs every problem to its own termination (converged, mu >= mu_max or
 * max_iter iterations); the host is not involved per iteration.  ref layout.
 *   in : x0 [B,n], us_init [B,T,m]
 *   io : delta [B]   (the solver's delta_, persists across calls; 2.0 at construction)
 *        k [B,T,m], K [B,T,m,n] (the solver's k_/K_; zeros at construction)
 *   out: xs [B,T+1,n], us [B,T,m], J [B], mu [B], iters [B], status [B] (ILQR_RAN_OUT..)
 *   trace: NULL or device arrays as described at ilqr_trace.
 *   workspace: device buffer of at least ilqr_solve_workspace_bytes(model, B).
 * The launch shape (problems per wavefront, wavefronts per workgroup, and how many wavefronts share the
 * backward sweep of one group of problems) is chosen from B, T and max_iter and never changes a result; for
 * experiments the environment variables ILQR_SOLVE_GROUP (1, 2 or 3 wavefronts per group), ILQR_SOLVE_LPW
 * (16, 32 or 64 problems per group) and ILQR_SOLVE_BLOCK (64, 128 or 256 threads; only with one wavefront
 * per group -- a group of two or three wavefronts IS the workgroup, and the variable is ignored) override it
 * (every combination is covered by the launch-shape parity test).  They are read when the launch is issued; a launch recorded in a HIP graph keeps the
 * shape it was captured with. */
size_t ilqr_solve_workspace_bytes(const ilqr_model* model, int64_t B);
int ilqr_solve_batched(const ilqr_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                       const double* us_init, double* delta, double* k, double* K, double* xs,
                       double* us, double* J, double* mu, int32_t* iters, int32_t* status,
                       const ilqr_trace* trace, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ilqr_solve_batched for callers that cannot allocate device memory and have no workspace to pass: code
 * under HIP graph capture, tight control loops, harnesses that forbid hipMalloc.  No counterpart in the
 * reference.  Arguments and results as ilqr_solve_batched without the workspace: the ping-pong copies of
 * the trajectories and gains live in a buffer of ILQR_STATIC_WORKSPACE_BYTES that belongs to the library
 * (one per GPU, held until the process ends), and the call allocates and frees nothing.  Returns
 * ILQR_ERR_WORKSPACE when ilqr_solve_workspace_bytes(model, B) exceeds that size (32 MiB: e.g. 11 000 problems
 * at T = 15, 1 700 at T = 100).  All calls on one GPU share the buffer: calls that may overlap on one GPU
 * (different streams, different threads) race -- serialise them, or use ilqr_solve_batched with workspaces of
 * their own.  The first call on a device looks the buffer's address up (hipGetSymbolAddress); make it
 * outside a graph capture. */
#define ILQR_STATIC_WORKSPACE_BYTES (32u << 20)
int ilqr_solve_batched_static(const ilqr_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                              const double* us_init, double* delta, double* k, double* K, double* xs,
                              double* us, double* J, double* mu, int32_t* iters, int32_t* status,
                              const ilqr_trace* trace, void* stream);

/* The same solve for batches that fill the GPU, cut into launches of `chunk` iterations: after each
 * launch the problems whose update_u loop has ended (converged, mu >= mu_max: ilqr_core.py:325-339)
 * are dropped and the rest continues as a dense sub-batch from its carried state (controls, mu,
 * delta, k, K and the Delta_J sums of its last backward pass), so that finished problems stop costing
 * lanes and memory traffic.  Results are those of ilqr_solve_batched (a continued launch re-rolls the
 * current controls: same inputs, same bits).  opts->dJ_carry is ignored (the sums live in the workspace).
 * No counterpart in the reference, which solves one problem at a time.  Arguments as
 * ilqr_solve_batched without the trace; the workspace is slightly larger (the index list of the running
 * problems and compact copies of their scalars; trajectories and gains are converted in place through
 * the index list); us may alias us_init.  Synchronises the stream once per chunk (the next launch is sized
 * on the host), so it is not capturable in a HIP graph. */
size_t ilqr_solve_chunked_workspace_bytes(const ilqr_model* model, int64_t B);
int ilqr_solve_batched_chunked(const ilqr_model* model, const ilqr_opts* opts, int32_t chunk, int64_t B,
                               const double* x0, const double* us_init, double* delta, double* k, double* K,
                               double* xs, double* us, double* J, double* mu, int32_t* iters, int32_t* status,
                               void* workspace, size_t workspace_bytes, void* stream);

/* CartPoleContinuousEnv._state_eq (ilqr/cartpole_gym.py:60-72), the plant the MPC
 * example steps: st [B,4] = [x, x', theta, theta'], u [B] -> out [B,4].
 * par HOST [5] = gravity, masscart, masspole, length, tau. */
int ilqr_cartpole_plant_step(int64_t B, const double* par, const double* st, const double* u,
                             double* out, void* stream);

/* Everything between two solves of the closed loop of examples/cartpole_example_mpc.py:66-79 (env.step,
 * set_state) around RecedingHorizonController.control (ilqr/ilqr_core.py:451-465), in one launch:
 *   u_applied [B]    = us_solved[:, 0]                          the control handed to the plant (:457-459)
 *   st [B,4]         = _state_eq(st, u_applied)                 in place (cartpole_gym.py:60-72)
 *   x0_aug [B,5]     = augment_state(st)                        for the next solve (cartpole.py:97-116)
 *   us_seed [B,N]    = us_solved moved one step along, its last control repeated (:463-465, step_size 1)
 * us_seed must not alias us_solved.  par as for ilqr_cartpole_plant_step.  Same arithmetic, bit for bit, as
 * the separate calls. */
int ilqr_cartpole_mpc_advance(int64_t B, int64_t N, const double* par, double* st, const double* us_solved,
                              double* us_seed, double* u_applied, double* x0_aug, void* stream);

/* Diagnostic hook (no counterpart in the reference): with a device buffer of
 * 4 * ceil(B/64) uint64 set, every wavefront of the next ilqr_solve_batched launches
 * records [start, end] of its life in 100 MHz ticks, its HW_ID and its XCC id.
 * NULL switches the log off.  Synchronous (hipMemcpyToSymbol). */
int ilqr_debug_wave_log(void* device_buffer);

/* Test hook: the library's own sin/cos/atan2 on arrays (a, b) -> sin a, cos a, atan2(a, b). */
int ilqr_detmath(int64_t n, const double* a, const double* b, double* sin_a, double* cos_a,
                 double* atan2_ab, double* sqrt_abs_a, double* a_over_b, void* stream);

/* ---- large-state path: n = 32 states, m = 8 controls, one problem per wavefront ----
 *
 * The reference ships no model of this size; the path exists for its generic machinery
 * (the same update_u / _backward_pass / boxQP / _forward_pass, ilqr_core.py:85-399 and
 * ilqr_boxQP.py:9-186, with dense 32 x 32 and 8 x 8 matrices) and runs the synthetic model
 *     x+ = x + dt (A x + B u + c phi(W x)),   phi(z) = z / sqrt(1 + z^2) element-wise,
 * with a QRCost (ilqr/ilqr_cost.py:490-632) and box limits on u.  Arrays are in the
 * reference's own layout ([B,T+1,n], [B,T,m], [B,T,m,n]); no batch-minor conversion.
 * Every pointer of this struct is a DEVICE pointer (the matrices are read by the kernels). */
typedef struct ilqr_large_model {
    int32_t n, m, T;  /* 1 <= n <= 32 states, 1 <= m <= 8 controls.  The kernels work on 32 x 8: every array of
                         this struct and of the entry points keeps the full-size layout ([32*32], [32*8], [8*8],
                         [32], [8]; trajectories [B,T+1,32], [B,T,8], gains [B,T,8,32]) with the entries beyond n
                         and m ZERO (limits of a padded control: lower = upper = 0).  Padding adds exact zeros to
                         the end of every sum of the reference's arithmetic, so the first n / m entries of every
                         result equal the n x m computation bit for bit, and boxQP treats the padded controls as
                         variables that do not exist (result codes, free sets and counters are those of m). */
    int32_t cost_kind; /* ILQR_COST_QR, or ILQR_COST_PATH_QR: x_goal [T+1][n], u_goal [T][m] hold the goals
                          of every time step (PathQRCost, ilqr/ilqr_cost.py:635-770) */
    double dt, sig_c; /* step and the gain c of the non-linearity */
    const double *A, *B, *W;          /* [n*n], [n*m], [n*n] row-major */
    const double *Q, *R, *Q_terminal; /* [n*n], [m*m], [n*n] */
    const double *x_goal, *u_goal;    /* [n], [m] (or the paths, see cost_kind) */
    const double *lower, *upper;      /* [m] */
    /* 0, 0: the goals above are shared by the batch.  Non-zero (both): problem b has its own goal (or path) at
     * x_goal + b * goal_stride_x and u_goal + b * goal_stride_u (doubles); accepted by every ilqr_large_* entry
     * point.  See ilqr_model for the counterpart in the reference. */
    int64_t goal_stride_x, goal_stride_u;
} ilqr_large_model;

/* iLQR._dynamics_rollout + _trajectory_cost (ilqr_core.py:69-83, :230-239):
 * x0 [B,n], us [B,T,m] -> xs [B,T+1,n], J [B]. */
int ilqr_large_rollout_cost(const ilqr_large_model* model, int64_t B, const double* x0, const double* us,
                            double* xs, double* J, void* stream);

/* Dynamics.f_x / f_u (ilqr_dynamics.py:498-524) and Cost.l_x, l_u, l_xx, l_uu, l_ux (ilqr_cost.py:555-632) of
 * the synthetic model for every (problem, step), plus the terminal l_x, l_xx: the tensors _compute_Q_terms
 * (ilqr_core.py:189-198) asks for, as dense streams in the reference's layout (the struct of ilqr_deriv_bm,
 * here batch outermost):
 *   fx [B,T,n,n]  fu [B,T,n,m]  lx [B,T,n]  lu [B,T,m]  lxx [B,T,n,n]  luu [B,T,m,m]  lux [B,T,m,n]
 *   lxT [B,n]  lxxT [B,n,n]            (21.3 KB per problem and step).
 * xs [B,T+1,n], us [B,T,m].  The solve kernels evaluate these in registers instead; this entry point exists
 * for callers that want the tensors themselves and to measure derivative streaming. */
int ilqr_large_deriv(const ilqr_large_model* model, int64_t B, const double* xs, const double* us,
                     const ilqr_deriv_bm_t* out, void* stream);

/* iLQR._backward_pass (ilqr_core.py:109-228) with boxQP: xs, us, mu [B] -> k [B,T,m],
 * K [B,T,m,n], dJ [2,B], info [5,B] (as ilqr_backward_bm with ld = B); optional
 * qp_result [B,T] and free_mask [B,T,m]. */
int ilqr_large_backward(const ilqr_large_model* model, const ilqr_opts* opts, int64_t B, const double* xs,
                        const double* us, const double* mu, double* k, double* K, double* dJ, int32_t* info,
                        int32_t* qp_result, int32_t* free_mask, void* stream);

/* iLQR._forward_pass + _trajectory_cost (ilqr_core.py:85-107) for n_alpha step sizes (HOST array),
 * 16 at a time as the columns of one matrix: J_new [B, n_alpha]; the trajectory of
 * alphas[store_idx] goes to xs_new [B,T+1,n], us_new [B,T,m] (store_idx < 0: none). */
int ilqr_large_forward_ls(const ilqr_large_model* model, int64_t B, int32_t n_alpha, const double* alphas,
                          int32_t store_idx, const double* xs, const double* us, const double* k, const double* K,
                          double* J_new, double* xs_new, double* us_new, void* stream);

/* Diagnostic hook (no counterpart in the reference): with a device buffer of 8 * B uint64 set,
 * every backward sweep records per problem the shader-clock totals of its three phases (matrix
 * part 1, boxQP and gains, matrix part 2), T, and a breakdown of the second phase.  NULL switches it off.  Synchronous. */
int ilqr_large_debug_profile(void* device_buffer);

/* iLQR.update_u (ilqr_core.py:241-399), arguments as ilqr_solve_batched.  One problem per wavefront; the
 * wavefronts of at most one workgroup per CU take problems from a queue (a counter at the end of the
 * workspace, zeroed on the stream before the launch), so a long solve does not hold up its neighbours. */
size_t ilqr_large_workspace_bytes(const ilqr_large_model* model, int64_t B);
int ilqr_large_solve_batched(const ilqr_large_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                             const double* us_init, double* delta, double* k, double* K, double* xs, double* us,
                             double* J, double* mu, int32_t* iters, int32_t* status, const ilqr_trace* trace,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---- wide path: up to 64 states and 16 controls, one problem per workgroup, matrices resident in LDS ----
 *
 * The reference's sweep is dimension-agnostic (ilqr_core.py:109-228); these entry points run it for models too
 * large for a wavefront's registers: the same struct, the same synthetic model family and cost kinds, the same
 * per-problem goals, with 1 <= n <= 64 and 1 <= m <= 16.  DENSE layout: every array has the model's own size
 * (A [n*n], B [n*m], R [m*m], x_goal [n], ...; trajectories [B,T+1,n], [B,T,m]; gains [B,T,m,n]) -- no padding.
 * Semantics and outputs of each call are those of its ilqr_large_* namesake; results are bit-identical to it
 * where both apply (n <= 32, m <= 8). */
int ilqr_xl_rollout_cost(const ilqr_large_model* model, int64_t B, const double* x0, const double* us, double* xs,
                         double* J, void* stream);
int ilqr_xl_backward(const ilqr_large_model* model, const ilqr_opts* opts, int64_t B, const double* xs,
                     const double* us, const double* mu, double* k, double* K, double* dJ, int32_t* info,
                     int32_t* qp_result, int32_t* free_mask, void* stream);
int ilqr_xl_forward_ls(const ilqr_large_model* model, int64_t B, int32_t n_alpha, const double* alphas,
                       int32_t store_idx, const double* xs, const double* us, const double* k, const double* K,
                       double* J_new, double* xs_new, double* us_new, void* stream);
/* iLQR.update_u (ilqr_core.py:241-399), arguments as ilqr_large_solve_batched.  The workspace holds, per resident
 * workgroup (one per CU, or B if fewer), four candidate trajectories of the line search (four step sizes are
 * rolled out at a time, one per wavefront, and scanned in the reference's order) and a second copy of the gains. */
size_t ilqr_xl_workspace_bytes(const ilqr_large_model* model, int64_t B);
int ilqr_xl_solve_batched(const ilqr_large_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                          const double* us_init, double* delta, double* k, double* K, double* xs, double* us,
                          double* J, double* mu, int32_t* iters, int32_t* status, const ilqr_trace* trace,
                          void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ILQR_HIP_H */
