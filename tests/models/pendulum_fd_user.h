/* The damped pendulum of pendulum_user.h with NO analytic Jacobian: f_x, f_u by forward
 * differences, as the reference's FiniteDifferenceDynamics computes them. */
#define ILQR_USER_N 3
#define ILQR_USER_NP 4
#define ILQR_USER_FD

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
