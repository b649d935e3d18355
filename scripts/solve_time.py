"""Time of one whole solve of the benchmark batch with a given build of the library (A/B comparisons on one box):
python scripts/solve_time.py <libilqr_hip.so> [problems] [repetitions]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip  # noqa: E402
import bench  # noqa: E402

_hip.LIB_PATH = os.path.abspath(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
Q, R, Qt, xg = bench.cartpole_cost()
M = _hip.Model(bench.T, Q, R, Qt, xg, None, [-5.0], [5.0])
x0_h, us_h = bench.make_batch(B, 0)
x0, us = torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev)
ws = None
ms = []
for r in range(reps + 1):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    res = _hip.solve(M, _hip.Opts(), x0, us, workspace=ws)
    b.record()
    torch.cuda.synchronize()
    ws = res["workspace"]
    if r > 0:
        ms.append(a.elapsed_time(b))
print("%s B=%d group=%s: min %.2f ms, median %.2f ms, J checksum %.17g" % (
    os.path.basename(sys.argv[1]), B, os.environ.get("ILQR_SOLVE_GROUP", "-"), min(ms), float(np.median(ms)),
    float(res["J"].sum())))
