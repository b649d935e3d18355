"""How often the line search goes past alpha_0, by iteration index, on the benchmark batch (from the solve's trace):
python scripts/linesearch_profile.py [B=16384] [lanes per wave=32]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
lpw = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
Q, R, Qt, xg = bench.cartpole_cost()
M = _hip.Model(100, Q, R, Qt, xg, None, [-5.0], [5.0])
x0_h, us_h = bench.make_batch(B, 0)
res = _hip.solve(M, _hip.Opts(), torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev), trace=True)
it = res["iters"].cpu().numpy()
tr = {k: v.cpu().numpy() for k, v in res["trace"].items()}
n_it = tr["accepted"].shape[0]
nw = B // lpw
active = (np.arange(n_it)[:, None] < it[None, :])
aidx, acc = tr["alpha_idx"], tr["accepted"]
beyond = active & (aidx >= 1)
redo = active & (acc == 1) & (aidx >= 1)
rej = active & (acc == 0)
bw = beyond[:, :nw * lpw].reshape(n_it, nw, lpw)
aw = active[:, :nw * lpw].reshape(n_it, nw, lpw)
rw = redo[:, :nw * lpw].reshape(n_it, nw, lpw)
print("iterations  running  lanes past alpha_0  rejected  waves(of running) with a pass  passes/wave  waves with a re-run")
for lo, hi in ((0, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 200)):
    sl = slice(lo, hi)
    run_w = aw[sl].any(2)
    print("  %3d-%3d  %8.0f  %17.2f%%  %7.2f%%  %28.1f%%  %11.2f  %18.1f%%" % (
        lo, hi, active[sl].sum(1).mean(), 100.0 * beyond[sl].sum() / max(active[sl].sum(), 1),
        100.0 * rej[sl].sum() / max(active[sl].sum(), 1),
        100.0 * bw[sl].any(2).sum() / max(run_w.sum(), 1),
        np.ceil(bw[sl].sum(2) / 8.0).sum() / max(run_w.sum(), 1),
        100.0 * rw[sl].any(2).sum() / max(run_w.sum(), 1)))
