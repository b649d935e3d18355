"""Diagnostic: which features of a wave (its problems' iteration counts, line-search rejections, boxQP work,
the XCD it ran on) go with its lifetime in the solve kernel: python scripts/wave_correlates.py [B=65536]
(run with ILQR_SOLVE_LPW=64 below 32769 problems: the script assumes 64 problems per wave)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
Q, R, Qt, xg = bench.cartpole_cost()
M = _hip.Model(100, Q, R, Qt, xg, None, [-5.0], [5.0])
x0_h, us_h = bench.make_batch(B, 0)
x0 = torch.tensor(x0_h, device=dev); us = torch.tensor(us_h, device=dev)
nw = B // 64
O = _hip.Opts()
_hip.solve(M, O, x0, us); torch.cuda.synchronize()
lifes = []
for rep in range(2):
    log = torch.zeros((nw, 4), dtype=torch.int64, device=dev)
    _hip.check(_hip.lib().ilqr_debug_wave_log(C.c_void_p(log.data_ptr())))
    res = _hip.solve(M, O, x0, us, trace=(rep == 1)); torch.cuda.synchronize()
    _hip.check(_hip.lib().ilqr_debug_wave_log(None))
    L = log.cpu().numpy()
    lifes.append((L[:, 1] - L[:, 0]) / 1e5)
    xcc = L[:, 3]
lifes = np.array(lifes)
print("run-to-run correlation of per-wave life (second run writes the trace):", np.corrcoef(lifes)[0, 1])
life = lifes[0]
it = res["iters"].cpu().numpy().reshape(nw, 64)
tr = {k: v.cpu().numpy() for k, v in res["trace"].items()}   # [max_iter, B]
n_it = tr["accepted"].shape[0]
active = (np.arange(n_it)[:, None, None] < it[None]).astype(bool)           # [it, wave, lane]
acc = tr["accepted"].reshape(n_it, nw, 64); aidx = tr["alpha_idx"].reshape(n_it, nw, 64)
qpb = tr["qp_backtrack"].reshape(n_it, nw, 64); qpi = tr["qp_iter"].reshape(n_it, nw, 64)
beyond0 = active & (aidx >= 1)          # lanes whose search went past alpha_0 (cooperative pass needed)
redo = active & (acc == 1) & (aidx >= 1)  # accepted at a later alpha: one more rollout for the whole wave
feat = {
    "sum iters": it.sum(1), "lanes at max iterations": (it == n_it).sum(1),
    "iterations with a search past alpha_0": beyond0.any(2).sum(0),
    "cooperative passes (8 problems each)": np.ceil(beyond0.sum(2) / 8.0).sum(0),
    "iterations with a re-run winner": redo.any(2).sum(0),
    "max-lane qp_backtrack, summed": np.where(active, qpb, 0).max(2).sum(0),
    "max-lane qp_iter, summed": np.where(active, qpi, 0).max(2).sum(0),
    "odd xcc": xcc % 2,
}
for k, v in feat.items():
    print("%-42s corr with life %5.2f   (mean %.1f)" % (k, np.corrcoef(life, v)[0, 1], np.mean(v)))
A_ = np.stack([np.ones(nw)] + [feat[k] for k in ("cooperative passes (8 problems each)", "iterations with a re-run winner",
                                                  "max-lane qp_backtrack, summed", "odd xcc")], 1).astype(float)
coef, *_ = np.linalg.lstsq(A_, life, rcond=None)
print("least squares: life = %.1f ms + %.3f ms/pass + %.3f ms/re-run + %.4f ms/backtrack + %.1f ms if odd xcc; residual rms %.2f ms"
      % (coef[0], coef[1], coef[2], coef[3], coef[4], np.sqrt(np.mean((A_ @ coef - life) ** 2))))
print("slowest waves:")
for w in np.argsort(-life)[:6]:
    print("  wave %4d life %.1f ms (other run %.1f) xcc %d:" % (w, life[w], lifes[1][w], xcc[w]),
          ", ".join("%s %d" % (k.split(" (")[0], feat[k][w]) for k in list(feat)[2:7]))
