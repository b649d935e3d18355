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
#include "../include/ilqr_user_fd.h"

int user_model_n(void) { return ILQR_USER_N; }
int user_model_np(void) { return ILQR_USER_NP; }
void user_model_f(const double *p, const double *x, double u, double *out) { ilqr_user_f(p, x, u, out); }
void user_model_jac(const double *p, const double *x, double u, double *fx, double *fu)
{
    ilqr_user_jac(p, x, u, fx, fu);
}
