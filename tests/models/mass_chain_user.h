/* Four masses on a line, each tied to its neighbours (and the first to a wall) by hardening springs, each with
 * its own actuator: state [p1..p4, v1..v4] (8 states), controls u = [f1..f4] (4 controls), parameters
 * p = [dt, k, k3, damping]:
 *     e_i = p_i - p_{i-1}  (p_0 = 0),   s(e) = k e + k3 e^3
 *     v_i' = u_i - s(e_i) + s(e_{i+1}) - damping v_i      (no spring beyond the last mass)
 * explicit Euler.  The largest model the per-model solver takes: 8 states and 4 controls. */
#define ILQR_USER_N 8
#define ILQR_USER_M 4
#define ILQR_USER_NP 4

ILQR_FN void ilqr_user_f(const double* p, const double* x, const double* u, double* out)
{
    double dt = p[0], k = p[1], k3 = p[2], d = p[3];
    double s[5];
    for (int i = 0; i < 4; ++i) {
        double e = x[i] - (i > 0 ? x[i - 1] : 0.0);
        s[i] = k * e + k3 * ((e * e) * e);
    }
    s[4] = 0.0;
    for (int i = 0; i < 4; ++i) {
        double a = ((u[i] - s[i]) + s[i + 1]) - d * x[4 + i];
        out[i] = x[i] + x[4 + i] * dt;
        out[4 + i] = x[4 + i] + a * dt;
    }
}

ILQR_FN void ilqr_user_jac(const double* p, const double* x, const double* u, double* fx, double* fu)
{
    double dt = p[0], k = p[1], k3 = p[2], d = p[3];
    (void)u;
    double ds[5]; /* s'(e_i) */
    for (int i = 0; i < 4; ++i) {
        double e = x[i] - (i > 0 ? x[i - 1] : 0.0);
        ds[i] = k + 3.0 * k3 * (e * e);
    }
    ds[4] = 0.0;
    for (int i = 0; i < 64; ++i)
        fx[i] = 0.0;
    for (int i = 0; i < 32; ++i)
        fu[i] = 0.0;
    for (int i = 0; i < 4; ++i) {
        fx[i * 8 + i] = 1.0;
        fx[i * 8 + 4 + i] = dt;
        /* a_i = u_i - s(p_i - p_{i-1}) + s(p_{i+1} - p_i) - d v_i */
        fx[(4 + i) * 8 + i] = -((ds[i] + ds[i + 1]) * dt);
        if (i > 0)
            fx[(4 + i) * 8 + i - 1] = ds[i] * dt;
        if (i < 3)
            fx[(4 + i) * 8 + i + 1] = ds[i + 1] * dt;
        fx[(4 + i) * 8 + 4 + i] = 1.0 - d * dt;
        fu[(4 + i) * 4 + i] = dt;
    }
}
