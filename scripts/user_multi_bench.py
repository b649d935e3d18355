"""Timing of the general (multi-control) path of the per-model solver on the planar rotor test
model (7 states, 2 controls): python scripts/user_multi_bench.py [B=16384] [T=50]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_amd import _hip, user_model as um  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
r = np.random.RandomState(7)
phi = r.uniform(-1.0, 1.0, B)
x0 = np.stack([r.uniform(-2, 2, B), r.uniform(-2, 2, B), np.sin(phi), np.cos(phi), r.uniform(-1, 1, B),
               r.uniform(-1, 1, B), r.uniform(-1, 1, B)], axis=1)
us0 = r.uniform(-1, 1, (B, T, 2))
M = um.UserModel(um.load("planar_rotor_user"), T, [0.05, 9.81, 1.0, 4.0, 0.1], np.diag([10.0, 10, 5, 5, 1, 1, 1]),
                 [[0.2, 0.05], [0.05, 0.2]], 50.0 * np.eye(7), [0, 0, 0, 1.0, 0, 0, 0], None, [-3.0, -3.0], [3.0, 3.0])
dx, du = torch.tensor(x0, device="cuda:0"), torch.tensor(us0, device="cuda:0")
O = _hip.Opts(max_iter=100)
res = um.solve(M, O, dx, du)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    um.solve(M, O, dx, du, workspace=res["workspace"])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
it = res["iters"].cpu().numpy()
print("planar rotor (n=7, m=2, T=%d), %d problems: %.1f ms per batch, %.0f solves/s, mean %.1f iterations (max %d), "
      "converged %.2f" % (T, B, dt * 1e3, B / dt, it.mean(), it.max(), float((res["status"] == 1).float().mean())))
