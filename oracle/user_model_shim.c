/* TEST INFRASTRUCTURE (CPU oracle): a user model header compiled for the CPU, so that the oracle
 * can run the same source the device library is built from.
 *   gcc $(CFLAGS) -shared -DILQR_USER_MODEL='"model.h"' user_model_shim.c -o libusermodel_<name>.so */
#include <math.h>

#include "det_math.h"

#define ILQR_FN static inline
#define ilqr_sincos(x, s, c) dm_sincos((x), (s), (c))
#define ilqr_atan2(y, x) dm_atan2((y), (x))
#define ilqr_sqrt(x) sqrt(x)
#include ILQR_USER_MODEL
#ifndef ILQR_USER_M
#define ILQR_USER_M 1
#endif
#include "../include/ilqr_user_fd.h"

int user_model_n(void) { return ILQR_USER_N; }
int user_model_m(void) { return ILQR_USER_M; }
int user_model_np(void) { return ILQR_USER_NP; }
/* u is always passed as a pointer here; a single-control model takes it by value */
void user_model_f(const double *p, const double *x, const double *u, double *out)
{
#if ILQR_USER_M > 1
    ilqr_user_f(p, x, u, out);
#else
    ilqr_user_f(p, x, u[0], out);
#endif
}
void user_model_jac(const double *p, const double *x, const double *u, double *fx, double *fu)
{
#if ILQR_USER_M > 1
    ilqr_user_jac(p, x, u, fx, fu);
#else
    ilqr_user_jac(p, x, u[0], fx, fu);
#endif
}
