/* Finite-difference Jacobian for a user model that supplies only ilqr_user_f: the counterpart of
 * the reference's FiniteDifferenceDynamics (ilqr/ilqr_dynamics.py:93-167), whose f_x / f_u are
 * scipy.optimize.approx_fprime of every output, i.e. forward differences with a fixed increment,
 *     f_x[m][k] = (f(x + x_eps e_k, u)[m] - f(x, u)[m]) / x_eps ,   f_u likewise with u_eps,
 * both increments sqrt(machine epsilon) = 2^-26 by default (:112-113).
 * A model header selects it with   #define ILQR_USER_FD   (optionally ILQR_USER_FD_XEPS /
 * ILQR_USER_FD_UEPS) instead of defining ilqr_user_jac; this file is included after the model
 * header by the device translation unit and by the CPU build of the same model. */
#ifndef ILQR_USER_FD_H
#define ILQR_USER_FD_H
#ifdef ILQR_USER_FD

#ifndef ILQR_USER_FD_XEPS
#define ILQR_USER_FD_XEPS 1.4901161193847656e-08
#endif
#ifndef ILQR_USER_FD_UEPS
#define ILQR_USER_FD_UEPS 1.4901161193847656e-08
#endif

#if defined(ILQR_USER_M) && ILQR_USER_M > 1
ILQR_FN void ilqr_user_jac(const double* p, const double* x, const double* u, double* fx, double* fu)
{
    double f0[ILQR_USER_N], f1[ILQR_USER_N], xk[ILQR_USER_N], uk[ILQR_USER_M];
    ilqr_user_f(p, x, u, f0);
    for (int k = 0; k < ILQR_USER_N; ++k) {
        for (int a = 0; a < ILQR_USER_N; ++a)
            xk[a] = x[a];
        xk[k] = x[k] + ILQR_USER_FD_XEPS;
        ilqr_user_f(p, xk, u, f1);
        for (int m = 0; m < ILQR_USER_N; ++m)
            fx[m * ILQR_USER_N + k] = (f1[m] - f0[m]) / ILQR_USER_FD_XEPS;
    }
    for (int k = 0; k < ILQR_USER_M; ++k) {
        for (int a = 0; a < ILQR_USER_M; ++a)
            uk[a] = u[a];
        uk[k] = u[k] + ILQR_USER_FD_UEPS;
        ilqr_user_f(p, x, uk, f1);
        for (int m = 0; m < ILQR_USER_N; ++m)
            fu[m * ILQR_USER_M + k] = (f1[m] - f0[m]) / ILQR_USER_FD_UEPS;
    }
}
#else
ILQR_FN void ilqr_user_jac(const double* p, const double* x, double u, double* fx, double* fu)
{
    double f0[ILQR_USER_N], f1[ILQR_USER_N], xk[ILQR_USER_N];
    ilqr_user_f(p, x, u, f0);
    for (int k = 0; k < ILQR_USER_N; ++k) {
        for (int a = 0; a < ILQR_USER_N; ++a)
            xk[a] = x[a];
        xk[k] = x[k] + ILQR_USER_FD_XEPS;
        ilqr_user_f(p, xk, u, f1);
        for (int m = 0; m < ILQR_USER_N; ++m)
            fx[m * ILQR_USER_N + k] = (f1[m] - f0[m]) / ILQR_USER_FD_XEPS;
    }
    ilqr_user_f(p, x, u + ILQR_USER_FD_UEPS, f1);
    for (int m = 0; m < ILQR_USER_N; ++m)
        fu[m] = (f1[m] - f0[m]) / ILQR_USER_FD_UEPS;
}

#endif

#endif /* ILQR_USER_FD */
#endif /* ILQR_USER_FD_H */
