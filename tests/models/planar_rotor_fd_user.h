/* The planar two-rotor vehicle of planar_rotor_user.h with NO analytic Jacobian: f_x, f_u by forward
 * differences, as the reference's FiniteDifferenceDynamics computes them -- here with two controls. */
#define ILQR_USER_N 7
#define ILQR_USER_M 2
#define ILQR_USER_NP 5
#define ILQR_USER_FD

ILQR_FN void ilqr_user_f(const double* p, const double* x, const double* u, double* out)
{
    double dt = p[0], g = p[1], im = p[2], ai = p[3], drag = p[4];
    double s = x[2], c = x[3], vx = x[4], vy = x[5], w = x[6];
    double a = im * (u[0] + u[1]) + g;
    double ax = -(a * s) - drag * vx;
    double ay = (a * c - g) - drag * vy;
    double aw = ai * (u[1] - u[0]);
    double phi = ilqr_atan2(s, c);
    double nphi = phi + w * dt;
    double sn, cn;
    ilqr_sincos(nphi, &sn, &cn);
    out[0] = x[0] + vx * dt;
    out[1] = x[1] + vy * dt;
    out[2] = sn;
    out[3] = cn;
    out[4] = vx + ax * dt;
    out[5] = vy + ay * dt;
    out[6] = w + aw * dt;
}
