/* C ABI of a per-model library for USER-DEFINED dynamics (up to 8 states, 1 to 4 controls).
 *
 * The reference accepts any Python object with f / f_x / f_u (AutogradDynamics,
 * FiniteDifferenceDynamics: ilqr/ilqr_dynamics.py:94-575) and calls it once per time step.
 * Callbacks cannot run inside a kernel, so here a model is a small C source ("model header")
 *
 *     #define ILQR_USER_N  <states, 1..8>
 *     #define ILQR_USER_NP <number of double parameters p>
 *     ILQR_FN void ilqr_user_f  (const double* p, const double* x, double u, double* out);
 *     ILQR_FN void ilqr_user_jac(const double* p, const double* x, double u,
 *                                double* fx, double* fu);   (fx: [N][N] row-major, fu: [N])
 *
 * (or, instead of ilqr_user_jac,  #define ILQR_USER_FD : forward-difference Jacobian as the
 * reference's FiniteDifferenceDynamics, see ilqr_user_fd.h).  With  #define ILQR_USER_M <2..4>
 * the model has that many controls and both functions take  const double* u  (fu: [N][M]
 * row-major); such models run on a general per-lane path (dense boxQP; line search as for one control).
 * The functions
 * may use ilqr_sincos(x, &s, &c), ilqr_atan2(y, x), ilqr_sqrt(x) and plain arithmetic;
 * ilqr_amd/csrc/ilqr_user_tu.hip is compiled around it into one shared library per model
 * (ilqr_amd.user_model.build_user_model does that with hipcc).  That library exports the
 * functions below; everything else (options, trace, array layouts, per-problem status)
 * is as for ilqr_solve_batched in ilqr_hip.h.
 *
 * Models with MORE than 8 states or 4 controls (up to 32 states and 8 controls) use the same
 * header contract but are compiled into the large-state path instead:
 * ilqr_amd/csrc/ilqr_large.hip with -DILQR_LARGE_USER_MODEL='"model.h"'
 * (ilqr_amd.user_model.build_large_user_model).  That library exports the ilqr_large_* entry
 * points declared in ilqr_hip.h -- ilqr_large_model.A then points to the model's parameter
 * vector (DEVICE, ILQR_USER_NP doubles), B and W are not used, n and m must equal the model's --
 * and  int ilqr_large_user_dims(int32_t* n, int32_t* m, int32_t* n_params).  ilqr_large_deriv
 * returns ILQR_ERR_UNSUPPORTED there. */
#ifndef ILQR_USER_H
#define ILQR_USER_H

#include <stddef.h>
#include <stdint.h>

#include "ilqr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Model parameters + QRCost (ilqr/ilqr_cost.py:490-530) + control bounds; HOST pointers. */
typedef struct ilqr_user_model {
    int32_t n, m, T;    /* n must equal the library's ILQR_USER_N, m its ILQR_USER_M (1 if not defined) */
    int32_t n_params;   /* must equal ILQR_USER_NP */
    int32_t cost_kind;  /* ILQR_COST_QR, or ILQR_COST_PATH_QR: x_goal / u_goal are then DEVICE arrays
                           [T+1][n] / [T][m] (PathQRCost, ilqr/ilqr_cost.py:635-770), both required */
    int32_t pad_;
    const double* params;             /* [n_params] */
    const double *Q, *R, *Q_terminal; /* [n*n], [m*m], [n*n] (Q_terminal NULL = Q) */
    const double *x_goal, *u_goal;    /* [n], [m] (NULL = zeros) */
    const double *lower, *upper;      /* [m] */
    /* 0, 0: shared goals.  Non-zero (both): x_goal / u_goal are DEVICE arrays with one goal (QR) or one path
     * (PATH_QR) per problem, problem b's at x_goal + b * goal_stride_x, u_goal + b * goal_stride_u (doubles).
     * See ilqr_model in ilqr_hip.h. */
    int64_t goal_stride_x, goal_stride_u;
} ilqr_user_model;

/* The dimensions the library was compiled for (states, parameters; controls). */
int ilqr_user_dims(int32_t* n, int32_t* n_params);
int ilqr_user_controls(void);

/* iLQR.update_u (ilqr_core.py:241-399) for B problems; arguments as ilqr_solve_batched. */
size_t ilqr_user_workspace_bytes(const ilqr_user_model* model, int64_t B);
int ilqr_user_solve_batched(const ilqr_user_model* model, const ilqr_opts* opts, int64_t B, const double* x0,
                            const double* us_init, double* delta, double* k, double* K, double* xs, double* us,
                            double* J, double* mu, int32_t* iters, int32_t* status, const ilqr_trace* trace,
                            void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ILQR_USER_H */
