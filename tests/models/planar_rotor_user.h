/* A planar two-rotor vehicle, state [px, py, sin(phi), cos(phi), vx, vy, w] (7 states), controls
 * u = [thrust_left, thrust_right] around hover, parameters p = [dt, g, inv_mass, arm_over_inertia, drag]:
 *     a = inv_mass (u0 + u1) + g            (total specific thrust; u = 0 hovers when level)
 *     vx' = -a sin(phi) - drag vx ,  vy' = a cos(phi) - g - drag vy ,  w' = arm_over_inertia (u1 - u0)
 * explicit Euler; the attitude is carried as (sin, cos) like the cart-pole's pole angle.
 * A model with TWO controls: it runs on the general path of the per-model solver. */
#define ILQR_USER_N 7
#define ILQR_USER_M 2
#define ILQR_USER_NP 5

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

ILQR_FN void ilqr_user_jac(const double* p, const double* x, const double* u, double* fx, double* fu)
{
    double dt = p[0], g = p[1], im = p[2], ai = p[3], drag = p[4];
    double s = x[2], c = x[3], w = x[6];
    double a = im * (u[0] + u[1]) + g;
    double r2 = s * s + c * c;
    double inv_r2 = 1.0 / r2;
    double th_s = c * inv_r2, th_c = -(s * inv_r2);
    double phi = ilqr_atan2(s, c);
    double nphi = phi + w * dt;
    double sn, cn;
    ilqr_sincos(nphi, &sn, &cn);
    for (int i = 0; i < 49; ++i)
        fx[i] = 0.0;
    for (int i = 0; i < 14; ++i)
        fu[i] = 0.0;
    fx[0 * 7 + 0] = 1.0;
    fx[0 * 7 + 4] = dt;
    fx[1 * 7 + 1] = 1.0;
    fx[1 * 7 + 5] = dt;
    fx[2 * 7 + 2] = cn * th_s;
    fx[2 * 7 + 3] = cn * th_c;
    fx[2 * 7 + 6] = cn * dt;
    fx[3 * 7 + 2] = -(sn * th_s);
    fx[3 * 7 + 3] = -(sn * th_c);
    fx[3 * 7 + 6] = -(sn * dt);
    fx[4 * 7 + 2] = -(a * dt);
    fx[4 * 7 + 4] = 1.0 - drag * dt;
    fx[5 * 7 + 3] = a * dt;
    fx[5 * 7 + 5] = 1.0 - drag * dt;
    fx[6 * 7 + 6] = 1.0;
    fu[4 * 2 + 0] = -(im * s) * dt;
    fu[4 * 2 + 1] = -(im * s) * dt;
    fu[5 * 2 + 0] = (im * c) * dt;
    fu[5 * 2 + 1] = (im * c) * dt;
    fu[6 * 2 + 0] = -(ai * dt);
    fu[6 * 2 + 1] = ai * dt;
}
