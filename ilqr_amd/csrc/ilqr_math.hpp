// fp64 sin/cos/atan2 for the device, built only from IEEE add/sub/mul/div, rint
// and floor so that results do not depend on the vendor math library: the
// same operation sequence evaluated on any IEEE machine gives the same bits.
// Compile with -ffp-contract=off; nothing here may be fused.
//
// These stand where the reference calls Theano's sin / cos / arctan2
// (ilqr/cartpole.py:83-91); they are within 1 ulp of glibc's.
//
// sin/cos: two-round Cody-Waite reduction by pi/2 (33-bit pieces, exact
// products for |x| < 2^20 * pi/2) and degree-13/14 minimax kernels on
// [-pi/4, pi/4].  atan2: four-breakpoint reduction and an 11-term odd
// polynomial.  Input magnitudes above 2^30 (and inf/nan) give NaN.
#pragma once
#include <hip/hip_runtime.h>

namespace ilqr {

__device__ __forceinline__ double ksin(double x, double y)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

__device__ __forceinline__ double kcos(double x, double y)
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

__device__ __forceinline__ void det_sincos(double x, double& sn, double& cs)
{
    const double PIO2_1 = 1.57079632673412561417e+00, PIO2_2 = 6.07710050630396597660e-11,
                 PIO2_2T = 2.02226624879595063154e-21, INVPIO2 = 6.36619772367581382433e-01;
    bool bad = !(fabs(x) <= 1073741824.0); // also inf / nan
    double q = rint(x * INVPIO2);
    double t = x - q * PIO2_1;
    double w = q * PIO2_2;
    double r = t - w;
    w = q * PIO2_2T - ((t - r) - w);
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double s = ksin(y0, y1);
    double c = kcos(y0, y1);
    double quad = q - 4.0 * floor(q * 0.25);
    // quadrant selection as selects (no divergent control flow)
    bool swap = (quad == 1.0) || (quad == 3.0);
    bool neg_s = (quad == 2.0) || (quad == 3.0);
    bool neg_c = (quad == 1.0) || (quad == 2.0);
    double ss = swap ? c : s;
    double cc = swap ? s : c;
    ss = neg_s ? -ss : ss;
    cc = neg_c ? -cc : cc;
    sn = bad ? __builtin_nan("") : ss;
    cs = bad ? __builtin_nan("") : cc;
}

__device__ __forceinline__ double det_atan_pos(double x)
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

__device__ __forceinline__ double det_atan2(double y, double x)
{
    const double PI = 3.1415926535897931160e+00, PI_LO = 1.2246467991473531772e-16;
    bool xneg = __builtin_signbit(x) != 0;
    bool yneg = __builtin_signbit(y) != 0;
    double ax = fabs(x), ay = fabs(y);
    // general case first, special cases override it (selects, no divergent branches)
    double z = det_atan_pos(ay / ax);
    z = (__builtin_isinf(ax) && __builtin_isinf(ay)) ? 0.25 * PI : z;
    double zl = z - PI_LO;
    double res = xneg ? (yneg ? zl - PI : PI - zl) : (yneg ? -z : z);
    res = (ax == 0.0) ? (yneg ? -0.5 * PI : 0.5 * PI) : res;
    res = (ay == 0.0) ? (xneg ? (yneg ? -PI : PI) : y) : res; // on the x axis: +-0 or +-pi
    res = (x != x || y != y) ? __builtin_nan("") : res;
    return res;
}

} // namespace ilqr
