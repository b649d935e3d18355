"""Timing of the per-model solver at its size limits, the mass chain test model (8 states, 4 controls):
python scripts/user_chain_bench.py [B=16384] [T=50]"""
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
r = np.random.RandomState(5)
x0 = np.concatenate([r.uniform(-0.8, 0.8, (B, 4)), r.uniform(-0.5, 0.5, (B, 4))], axis=1)
us0 = r.uniform(-1, 1, (B, T, 4))
M = um.UserModel(um.load("mass_chain_user"), T, [0.05, 4.0, 0.5, 0.5], np.diag([8.0] * 4 + [1.0] * 4),
                 0.1 * np.eye(4) + 0.02 * np.ones((4, 4)), 40.0 * np.eye(8), np.zeros(8), None, [-0.6] * 4, [0.6] * 4)
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
print("mass chain (n=8, m=4, T=%d), %d problems: %.1f ms per batch, %.0f solves/s, mean %.1f iterations (max %d), "
      "converged %.2f" % (T, B, dt * 1e3, B / dt, it.mean(), it.max(), float((res["status"] == 1).float().mean())))
