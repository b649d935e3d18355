"""Where the two wavefronts of a pair spend a backward sweep (diagnostic build -DILQR_PAIR_PROF):
python scripts/pair_profile.py build/libilqr_pairprof.so"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip  # noqa: E402
import bench  # noqa: E402

_hip.LIB_PATH = os.path.abspath(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = torch.device("cuda:0")
Q, R, Qt, xg = bench.cartpole_cost()
M = _hip.Model(bench.T, Q, R, Qt, xg, None, [-5.0], [5.0])
x0_h, us_h = bench.make_batch(B, 0)
x0, us = torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev)
L = _hip.lib()
buf = (C.c_ulonglong * 32)()
_hip.solve(M, _hip.Opts(), x0, us)
torch.cuda.synchronize()
L.ilqr_debug_pair_prof(buf, 1)
res = _hip.solve(M, _hip.Opts(), x0, us)
torch.cuda.synchronize()
L.ilqr_debug_pair_prof(buf, 0)
v = np.array(list(buf), dtype=np.float64).reshape(4, 8)
steps = 200.0 * bench.T  # workgroup 0 holds a problem that runs all 200 iterations (checked below)
print("iterations of the first 32 problems: max", int(res["iters"][:32].max()))
print("master per step (cycles): wait at #1 %.0f | Q_uu, Cholesky, guess %.0f | wait at #2 %.0f | verification %.0f"
      % tuple(v[0, :4] / steps))
print("master: wrong guesses in %.2f %% of the wave-steps" % (100.0 * v[0, 4] / steps))
print("helper per step (cycles): wait at #1 %.0f | rest of the value update, Q-terms, next Jacobian %.0f | wait at #2 %.0f | "
      "gains + six numbers %.0f" % tuple(v[1, :4] / steps))
print("master per iteration (cycles): backward sweep(s) %.0f | first rollout %.0f | further step sizes, schedule, trace %.0f"
      % tuple(v[0, 5:8] / 200.0))
print("wrong guesses by lane-steps: left in pass 0 %d | left in pass 1 before its search %d | a search exhausted %d | "
      "common shape or literal loop %d (of %d lane-steps)" % (tuple(v[1, 4:8]) + (int(32 * steps),)))
if v[2].any():
    print("verifier per step (cycles): wait at #1 %.0f | first boxQP pass %.0f | wait at #2 %.0f | other passes, comparison %.0f | "
          "wrong guesses %.2f %% of the wave-steps" % (tuple(v[2, :4] / steps) + (100.0 * v[2, 4] / steps,)))
if v[3].any():
    print("shared rollout per step (cycles): master %.0f of which waiting %.0f | helper %.0f of which waiting %.0f"
          % tuple(v[3, :4] / steps))
