/* Ten masses on a line, each tied to its neighbours (and the first to a wall) by hardening springs, every second
 * mass with an actuator: state [p1..p10, v1..v10] (20 states), controls u = [f1, f3, f5, f7, f9] (5 controls),
 * parameters p = [dt, k, k3, damping]:
 *     e_i = p_i - p_{i-1}  (p_0 = 0),   s(e) = k e + k3 e^3
 *     v_i' = (u on odd-numbered masses) - s(e_i) + s(e_{i+1}) - damping v_i      (no spring beyond the last mass)
 * explicit Euler.  A model for the large-state path (more than 8 states): the same contract as the smaller models. */
#define ILQR_USER_N 20
#define ILQR_USER_M 5
#define ILQR_USER_NP 4

ILQR_FN void ilqr_user_f(const double* p, const double* x, const double* u, double* out)
{
    double dt = p[0], k = p[1], k3 = p[2], d = p[3];
    double s[11];
    for (int i = 0; i < 10; ++i) {
        double e = x[i] - (i > 0 ? x[i - 1] : 0.0);
        s[i] = k * e + k3 * ((e * e) * e);
    }
    s[10] = 0.0;
    for (int i = 0; i < 10; ++i) {
        double f = (i % 2 == 0) ? u[i / 2] : 0.0;
        double a = ((f - s[i]) + s[i + 1]) - d * x[10 + i];
        out[i] = x[i] + x[10 + i] * dt;
        out[10 + i] = x[10 + i] + a * dt;
    }
}

ILQR_FN void ilqr_user_jac(const double* p, const double* x, const double* u, double* fx, double* fu)
{
    double dt = p[0], k = p[1], k3 = p[2], d = p[3];
    (void)u;
    double ds[11]; /* s'(e_i) */
    for (int i = 0; i < 10; ++i) {
        double e = x[i] - (i > 0 ? x[i - 1] : 0.0);
        ds[i] = k + 3.0 * k3 * (e * e);
    }
    ds[10] = 0.0;
    for (int i = 0; i < 400; ++i)
        fx[i] = 0.0;
    for (int i = 0; i < 100; ++i)
        fu[i] = 0.0;
    for (int i = 0; i < 10; ++i) {
        fx[i * 20 + i] = 1.0;
        fx[i * 20 + 10 + i] = dt;
        fx[(10 + i) * 20 + i] = -((ds[i] + ds[i + 1]) * dt);
        if (i > 0)
            fx[(10 + i) * 20 + i - 1] = ds[i] * dt;
        if (i < 9)
            fx[(10 + i) * 20 + i + 1] = ds[i + 1] * dt;
        fx[(10 + i) * 20 + 10 + i] = 1.0 - d * dt;
        if (i % 2 == 0)
            fu[(10 + i) * 5 + i / 2] = dt;
    }
}
