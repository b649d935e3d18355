"""Write rate of the large-state derivative stream (ilqr_large_deriv): python scripts/large_deriv_bench.py [problems]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip  # noqa: E402
from ilqr_amd.synthetic import SyntheticWorkload  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = 200
dev = torch.device("cuda:0")
wl = SyntheticWorkload(T=T)
M = wl.device_model(device=dev)
x0, us = wl.batch(B)
x0, us = torch.tensor(x0, device=dev), torch.tensor(us, device=dev)
xs, _ = _hip.large_rollout_cost(M, x0, us)
out = _hip.large_deriv(M, xs, us)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
a.record()
for _ in range(reps):
    _hip.large_deriv(M, xs, us, out=out)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
n, m = 32, 8
nbytes = 8.0 * B * (T * (2 * n * n + 2 * n * m + n + m + m * m) + n + n * n)
print("ilqr_large_deriv: %d problems x %d steps, %.1f GB written in %.3f ms = %.2f TB/s (%.0f %% of 8 TB/s)"
      % (B, T, nbytes / 1e9, ms, nbytes / ms / 1e9, 100 * nbytes / ms / 1e9 / 8.0))
