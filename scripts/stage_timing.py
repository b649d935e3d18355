"""Time the stage kernels separately (diagnostic, not part of the bench contract)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
if len(sys.argv) > 2:  # a differently compiled build of the library (diagnostics)
    _hip.LIB_PATH = os.path.abspath(sys.argv[2])
T = 100
dev = torch.device("cuda:0")
Q, R, Qt, xg = bench.cartpole_cost()
M = _hip.Model(T, Q, R, Qt, xg, None, [-5.0], [5.0])
O = _hip.Opts()
x0_h, us_h = bench.make_batch(B, 0)
x0 = _hip.pack_bm(torch.tensor(x0_h, device=dev))
us = _hip.pack_bm(torch.tensor(us_h, device=dev))
mu = torch.ones(B, dtype=torch.float64, device=dev)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

xs, J = _hip.rollout_cost_bm(M, B, x0, us)
print("rollout_cost  %.3f ms" % timeit(lambda: _hip.rollout_cost_bm(M, B, x0, us)))
bw = _hip.backward_bm(M, O, B, xs, us, mu, want_masks=False)
print("backward      %.3f ms" % timeit(lambda: _hip.backward_bm(M, O, B, xs, us, mu, want_masks=False)))
for na in (1, 4, 30):
    al = O.alphas[:na]
    print("forward x%-2d   %.3f ms" % (na, timeit(lambda: _hip.forward_ls_bm(M, B, al, xs, us, bw["k"], bw["K"]))))
t_d = timeit(lambda: _hip.deriv_bm(M, B, xs, us))
stream_bytes = B * (T * 67 + 30) * 8.0
print("deriv stream  %.3f ms  (%.2f TB/s written)" % (t_d, stream_bytes / t_d / 1e9))
dv = _hip.deriv_bm(M, B, xs, us)
t_s = timeit(lambda: _hip.backward_bm(M, O, B, None, us, mu, want_masks=False, deriv=dv))
print("backward, staged (reads the derivative streams) %.3f ms  (%.2f TB/s read)" % (t_s, stream_bytes / t_s / 1e9))
# a converged-ish trajectory: backward pass cost late in the solve
res = _hip.solve(M, O, torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev))
xs2 = _hip.pack_bm(res["xs"]); us2 = _hip.pack_bm(res["us"]); mu2 = res["mu"].contiguous()
print("backward (final trajectories) %.3f ms" % timeit(lambda: _hip.backward_bm(M, O, B, xs2, us2, mu2, want_masks=False)))
bw2 = _hip.backward_bm(M, O, B, xs2, us2, mu2, want_masks=False)
print("forward x1 (final)            %.3f ms" % timeit(lambda: _hip.forward_ls_bm(M, B, O.alphas[:1], xs2, us2, bw2["k"], bw2["K"])))
info = bw2["info"].cpu().numpy()[:, :B]
print("final-traj backward: mean qp_backtrack/step %.1f max %d" % (info[2].mean() / T, info[2].max()))
