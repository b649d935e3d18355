"""boxQP event counts per wave-step of one backward sweep (diagnostic build with -DILQR_QP_PROF):
python scripts/qp_profile.py B build/qpprof.so"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1])
_hip.LIB_PATH = os.path.abspath(sys.argv[2])
T = 100
dev = torch.device("cuda:0")
Q, R, Qt, xg = bench.cartpole_cost()
M = _hip.Model(T, Q, R, Qt, xg, None, [-5.0], [5.0])
O = _hip.Opts()
x0_h, us_h = bench.make_batch(B, 0)
L = _hip.lib()
NAMES = ["calls", "passes", "armijo blocks", "jump blocks", "re-evaluations", "while trips", "jump blocks in pass>=1", "-",
         "0 passes", "1 pass", "2 passes", "3 passes", "4 passes", "5 passes", "6 passes", "7+ passes"]


def sweep(tag, xs, us, mu):
    buf = (C.c_ulonglong * 16)()
    L.ilqr_debug_qp_prof(buf, 1)
    _hip.backward_bm(M, O, B, xs, us, mu, want_masks=False)
    torch.cuda.synchronize()
    L.ilqr_debug_qp_prof(buf, 0)
    calls = max(1, buf[0])
    print("== %s: %d wave-steps" % (tag, buf[0]))
    for n, v in zip(NAMES, buf):
        if n != "-":
            print("   %-24s %10d  %.3f per wave-step" % (n, v, v / calls))


x0 = _hip.pack_bm(torch.tensor(x0_h, device=dev))
us = _hip.pack_bm(torch.tensor(us_h, device=dev))
mu = torch.ones(B, dtype=torch.float64, device=dev)
xs, J = _hip.rollout_cost_bm(M, B, x0, us)
sweep("initial trajectories", xs, us, mu)
res = _hip.solve(M, O, torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev))
sweep("final trajectories", _hip.pack_bm(res["xs"]), _hip.pack_bm(res["us"]), res["mu"].contiguous())
