"""Where a backward step of the wide path (ilqr_xl_*) spends its time: shader clocks of wavefront 0 per phase,
diagnostic build -DILQR_XL_PROF.   python scripts/xl_phases.py build/xlprof.so [n=64] [m=16] [B=256] [T=200]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip  # noqa: E402
from ilqr_amd.synthetic import SyntheticWorkload  # noqa: E402

_hip.LIB_PATH = os.path.abspath(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = int(sys.argv[3]) if len(sys.argv) > 3 else 16
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
T = int(sys.argv[5]) if len(sys.argv) > 5 else 200
dev = torch.device("cuda:0")
wl = SyntheticWorkload(n=n, m=m, T=T)
M = _hip.LargeModel(T, wl.A, wl.B, wl.W, wl.Q, wl.R, wl.Qt, wl.x_goal, wl.u_goal, wl.lower, wl.upper, dt=wl.dt,
                    sig_c=wl.sig_c, wide=True)
x0_h, us_h = wl.batch(B)
x0, us = torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev)
O = _hip.Opts(max_iter=12)
L = _hip._large_lib(M)
buf = (C.c_ulonglong * 16)()
names = {0: "(1) M1, M2, Q_x, Q_u", 1: "(2) Q_ux, Q_uu", 2: "(3) until every wavefront is through: boxQP (wave 0) / Q_xx + staging",
         12: "    of which wave 0 (PD test, warm start, boxQP, gains)", 6: "      factorisations", 7: "      triangular solves", 8: "      Armijo rounds", 3: "(4) T1", 4: "(5) value update", 5: "(6) symmetrise"}


def report(tag, steps):
    L.ilqr_xl_debug_prof(buf, 0)
    v = np.array(list(buf), dtype=np.float64)
    print(tag, "(cycles per step and problem, %d problem-steps)" % steps)
    for i, nm in names.items():
        print("  %-72s %9.0f" % (nm, v[i] / steps))


xs, J = _hip.large_rollout_cost(M, x0, us)
mu = torch.ones(B, dtype=torch.float64, device=dev)
_hip.large_backward(M, O, xs, us, mu, want_masks=False)
torch.cuda.synchronize()
L.ilqr_xl_debug_prof(buf, 1)
_hip.large_backward(M, O, xs, us, mu, want_masks=False)
torch.cuda.synchronize()
report("backward sweep on the initial trajectories", B * T)
out = _hip.large_solve(M, _hip.Opts(max_iter=60), x0, us)
torch.cuda.synchronize()
mu2 = out["mu"].contiguous()
L.ilqr_xl_debug_prof(buf, 1)
_hip.large_backward(M, O, out["xs"], out["us"], mu2, want_masks=False)
torch.cuda.synchronize()
report("backward sweep on the trajectories after 60 iterations", B * T)
