/* A damped pendulum with a torque limit, state [sin(theta), cos(theta), theta'] (theta = 0 hangs
 * down), parameters p = [dt, g_over_l, damping, inv_inertia]:
 *     theta'' = -g_over_l sin(theta) - damping theta' + inv_inertia u ,  explicit Euler.
 * A model the library has no built-in code for. */
#define ILQR_USER_N 3
#define ILQR_USER_NP 4

ILQR_FN void ilqr_user_f(const double* p, const double* x, double u, double* out)
{
    double dt = p[0], gl = p[1], damp = p[2], ii = p[3];
    double s = x[0], c = x[1], w = x[2];
    double acc = (-(gl * s) - damp * w) + ii * u;
    double th = ilqr_atan2(s, c);
    double nth = th + w * dt;
    double sn, cn;
    ilqr_sincos(nth, &sn, &cn);
    out[0] = sn;
    out[1] = cn;
    out[2] = w + acc * dt;
}

ILQR_FN void ilqr_user_jac(const double* p, const double* x, double u, double* fx, double* fu)
{
    double dt = p[0], gl = p[1], damp = p[2], ii = p[3];
    double s = x[0], c = x[1], w = x[2];
    double r2 = s * s + c * c;
    double inv_r2 = 1.0 / r2;
    double th_s = c * inv_r2, th_c = -(s * inv_r2);
    double th = ilqr_atan2(s, c);
    double nth = th + w * dt;
    double sn, cn;
    ilqr_sincos(nth, &sn, &cn);
    (void)u;
    fx[0] = cn * th_s;
    fx[1] = cn * th_c;
    fx[2] = cn * dt;
    fx[3] = -(sn * th_s);
    fx[4] = -(sn * th_c);
    fx[5] = -(sn * dt);
    fx[6] = -(gl * dt);
    fx[7] = 0.0;
    fx[8] = 1.0 - damp * dt;
    fu[0] = 0.0;
    fu[1] = 0.0;
    fu[2] = ii * dt;
}
