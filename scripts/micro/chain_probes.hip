// Probe kernels for scripts/chain_depth.py: each computes one link of the cart-pole solver's loop-carried
// recurrences from values loaded from memory to values stored to memory, with the library's own device functions,
// so that the length of its longest chain of dependent instructions can be read off the compiled code.
#include <hip/hip_runtime.h>

#include "../../ilqr_amd/csrc/ilqr_device.hpp"

using namespace ilqr;

// rollout, angle recurrence: (sin, cos)(t) -> atan2 -> + rate dt -> sin/cos -> (sin, cos)(t+1)
extern "C" __global__ void probe_angle(CartpoleP mp, const double* in, double* out)
{
    in += threadIdx.x * 32; // (per-lane operands, as in the solver: vector instructions, selects instead of branches)
    out += threadIdx.x * 4;
    CartpoleC P = cartpole_consts(mp);
    double sn, cn;
    cartpole_f_angle(P, in[0], in[1], in[2], sn, cn);
    out[0] = sn;
    out[1] = cn;
}

// rollout, velocity recurrence: x(t) -> u = u + alpha k + K (x_new - x) -> accelerations -> velocities(t+1)
extern "C" __global__ void probe_velocity(CartpoleP mp, const double* in, double* out)
{
    in += threadIdx.x * 32;
    out += threadIdx.x * 4;
    CartpoleC P = cartpole_consts(mp);
    double x[NX] = {in[0], in[1], in[2], in[3], in[4]};
    // u_new = u + alpha k + K . (x_new - x)   (ilqr_core.py:104), K . dx as the first product then fma
    double acc = in[10] * (x[0] - in[5]);
#pragma unroll
    for (int j = 1; j < NX; ++j)
        acc = fma(in[10 + j], x[j] - in[5 + j], acc);
    double u = (in[15] + in[16]) + acc;
    double o0, o1, o4;
    cartpole_f_velocities(P, x, u, o0, o1, o4);
    out[0] = o0;
    out[1] = o1;
    out[2] = o4;
}

// backward sweep: Q_uu -> sqrt -> reciprocal (the 1 x 1 Cholesky factor and what the solves multiply by)
extern "C" __global__ void probe_factor(const double* in, double* out)
{
    in += threadIdx.x * 32;
    out += threadIdx.x * 4;
    double d = sqrt(in[0]);
    double r = 1.0 / d;
    out[0] = d;
    out[1] = r;
}

// backward sweep: the boxQP outcome by the shortest chain (what the other wavefronts wait for)
extern "C" __global__ void probe_guess(QPOptsP O, const double* tab, const double* in, double* out)
{
    in += threadIdx.x * 32;
    out += threadIdx.x * 4;
    double x;
    bool fr;
    boxqp1_guess(O, tab, in[0], in[1], in[2], in[3], in[4], in[5], x, fr);
    out[0] = fr ? x : -x;
}
