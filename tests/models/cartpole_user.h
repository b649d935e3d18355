/* The cart-pole of ilqr/cartpole.py:72-92 written as a user model: the same expressions, in the
 * same order, as the library's built-in model, so that both paths must give the same bits.
 * State [x, x', sin(theta), cos(theta), theta'], parameters p = [dt, mc, mp, l, g]. */
#define ILQR_USER_N 5
#define ILQR_USER_NP 5

ILQR_FN void ilqr_user_f(const double* p, const double* x, double F, double* out)
{
    double dt = p[0], mc = p[1], mp = p[2], l = p[3], g = p[4];
    double xd = x[1], s = x[2], c = x[3], thd = x[4];
    double M = mc + mp;
    double mpl = mp * l;
    double temp = (F + mpl * (thd * thd) * s) / M;
    double num = g * s - c * temp;
    double den = l * (4.0 / 3.0 - mp * (c * c) / M);
    double thdd = num / den;
    double xdd = temp - mpl * thdd * c / M;
    double th = ilqr_atan2(s, c);
    double nth = th + thd * dt;
    double sn, cn;
    ilqr_sincos(nth, &sn, &cn);
    out[0] = x[0] + xd * dt;
    out[1] = xd + xdd * dt;
    out[2] = sn;
    out[3] = cn;
    out[4] = thd + thdd * dt;
}

ILQR_FN void ilqr_user_jac(const double* p, const double* x, double F, double* fx, double* fu)
{
    double dt = p[0], mc = p[1], mp = p[2], l = p[3], g = p[4];
    double s = x[2], c = x[3], thd = x[4];
    double M = mc + mp;
    double mpl = mp * l;
    double invM = 1.0 / M;
    double temp = (F + mpl * (thd * thd) * s) * invM;
    double num = g * s - c * temp;
    double den = l * (4.0 / 3.0 - mp * (c * c) * invM);
    double inv_den = 1.0 / den;
    double thdd = num * inv_den;
    double temp_s = mpl * (thd * thd) * invM;
    double temp_thd = 2.0 * mpl * thd * s * invM;
    double temp_F = invM;
    double num_s = g - c * temp_s;
    double num_c = -temp;
    double num_thd = -(c * temp_thd);
    double num_F = -(c * temp_F);
    double den_c = -(2.0 * l * mp * c * invM);
    double thdd_s = num_s * inv_den;
    double thdd_c = (num_c - thdd * den_c) * inv_den;
    double thdd_thd = num_thd * inv_den;
    double thdd_F = num_F * inv_den;
    double mplM = mpl * invM;
    double xdd_s = temp_s - mplM * (thdd_s * c);
    double xdd_c = -(mplM * (thdd_c * c + thdd));
    double xdd_thd = temp_thd - mplM * (thdd_thd * c);
    double xdd_F = temp_F - mplM * (thdd_F * c);
    double r2 = s * s + c * c;
    double inv_r2 = 1.0 / r2;
    double th_s = c * inv_r2;
    double th_c = -(s * inv_r2);
    double th = ilqr_atan2(s, c);
    double nth = th + thd * dt;
    double sn, cn;
    ilqr_sincos(nth, &sn, &cn);
    for (int i = 0; i < 25; ++i)
        fx[i] = 0.0;
    fx[0 * 5 + 0] = 1.0;
    fx[0 * 5 + 1] = dt;
    fx[1 * 5 + 1] = 1.0;
    fx[1 * 5 + 2] = xdd_s * dt;
    fx[1 * 5 + 3] = xdd_c * dt;
    fx[1 * 5 + 4] = xdd_thd * dt;
    fx[2 * 5 + 2] = cn * th_s;
    fx[2 * 5 + 3] = cn * th_c;
    fx[2 * 5 + 4] = cn * dt;
    fx[3 * 5 + 2] = -(sn * th_s);
    fx[3 * 5 + 3] = -(sn * th_c);
    fx[3 * 5 + 4] = -(sn * dt);
    fx[4 * 5 + 2] = thdd_s * dt;
    fx[4 * 5 + 3] = thdd_c * dt;
    fx[4 * 5 + 4] = 1.0 + thdd_thd * dt;
    fu[0] = 0.0;
    fu[1] = xdd_F * dt;
    fu[2] = 0.0;
    fu[3] = 0.0;
    fu[4] = thdd_F * dt;
}
