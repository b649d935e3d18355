/*
 * TEST INFRASTRUCTURE (CPU oracle) -- not part of the shipped product path.
 *
 * Portable fp64 sin/cos/atan2 built only from IEEE-754 add, sub, mul, div,
 * rint and floor.  Every operation is correctly rounded on x86-64 (SSE2) and
 * on gfx950, so the same sequence of operations gives the same bits on both
 * as long as the compiler is told not to contract a*b+c into an fma
 * (-ffp-contract=off).  The oracle's "deterministic" math mode uses these in
 * place of glibc's sin/cos/atan2 so that a GPU result can be compared
 * bit-for-bit; the "libm" mode (what numpy calls in the reference,
 * ilqr/cartpole.py:83-91 via Theano's sin/cos/arctan2) differs from these by
 * at most 1 ulp per call.
 *
 * Algorithms: Cody-Waite three-constant reduction by pi/2 followed by the
 * classic degree-13/14 minimax kernels for sin/cos on [-pi/4, pi/4], and the
 * four-breakpoint atan reduction with an 11-term odd polynomial.  These are
 * the published fdlibm algorithms (Sun Microsystems, 1993), restated.
 */
#ifndef ILQR_ORACLE_DET_MATH_H
#define ILQR_ORACLE_DET_MATH_H

#include <math.h>

/* pi/2 split into 33-bit pieces: q*piece is exact for |q| < 2^20. */
#define DM_PIO2_1  1.57079632673412561417e+00
#define DM_PIO2_1T 6.07710050650619224932e-11
#define DM_PIO2_2  6.07710050630396597660e-11
#define DM_PIO2_2T 2.02226624879595063154e-21
#define DM_INVPIO2 6.36619772367581382433e-01

/* Largest |x| for which a result is produced; beyond it (and for inf/nan) the
 * result is NaN.  Accuracy is < 1 ulp for |x| < 2^20 * pi/2 and degrades
 * smoothly above that. */
#define DM_TRIG_MAX 1073741824.0

static inline double dm_ksin(double x, double y)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

static inline double dm_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

static inline void dm_sincos(double x, double *sn, double *cs)
{
    if (!(fabs(x) <= DM_TRIG_MAX)) {
        *sn = NAN;
        *cs = NAN;
        return;
    }
    double q = rint(x * DM_INVPIO2);
    /* two rounds of Cody-Waite: remainder r - w good to ~118 bits */
    double t = x - q * DM_PIO2_1;
    double w = q * DM_PIO2_2;
    double r = t - w;
    w = q * DM_PIO2_2T - ((t - r) - w);
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double s = dm_ksin(y0, y1);
    double c = dm_kcos(y0, y1);
    double quad = q - 4.0 * floor(q * 0.25); /* 0,1,2,3 exactly */
    if (quad == 0.0) {
        *sn = s;
        *cs = c;
    } else if (quad == 1.0) {
        *sn = c;
        *cs = -s;
    } else if (quad == 2.0) {
        *sn = -s;
        *cs = -c;
    } else {
        *sn = -c;
        *cs = s;
    }
}

/* atan for x >= 0 (finite or +inf). */
static inline double dm_atan_pos(double x)
{
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    /* Range selection without branches: t = num / den with (num, den, hi, lo)
     * picked by interval; the first interval uses den = 1 (x / 1 is exact) and
     * hi = lo = 0, for which the general formula below reduces to t - t*(s1+s2). */
    double num = x, den = 1.0, hi = 0.0, lo = 0.0;
    if (x >= 0.4375) {
        num = 2.0 * x - 1.0;
        den = 2.0 + x;
        hi = 4.63647609000806093515e-01;
        lo = 2.26987774529616870924e-17;
    }
    if (x >= 0.6875) {
        num = x - 1.0;
        den = x + 1.0;
        hi = 7.85398163397448278999e-01;
        lo = 3.06161699786838301793e-17;
    }
    if (x >= 1.1875) {
        num = x - 1.5;
        den = 1.0 + 1.5 * x;
        hi = 9.82793723247329054082e-01;
        lo = 1.39033110312309984516e-17;
    }
    if (x >= 2.4375) {
        num = -1.0;
        den = x;
        hi = 1.57079632679489655800e+00;
        lo = 6.12323399573676603587e-17;
    }
    double t = num / den;
    double z = t * t;
    double w = z * z;
    double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    return hi - ((t * (s1 + s2) - lo) - t);
}

static inline double dm_atan2(double y, double x)
{
    const double PI = 3.1415926535897931160e+00, PI_LO = 1.2246467991473531772e-16;
    if (x != x || y != y)
        return NAN;
    int xneg = signbit(x) != 0;
    int yneg = signbit(y) != 0;
    double ax = fabs(x), ay = fabs(y);
    if (ay == 0.0) { /* on the x axis */
        if (!xneg)
            return y; /* +-0 */
        return yneg ? -PI : PI;
    }
    if (ax == 0.0)
        return yneg ? -0.5 * PI : 0.5 * PI;
    double z;
    if (isinf(ax) && isinf(ay))
        z = 0.25 * PI;
    else
        z = dm_atan_pos(ay / ax);
    if (!xneg)
        return yneg ? -z : z;
    return yneg ? (z - PI_LO) - PI : PI - (z - PI_LO);
}

#endif
