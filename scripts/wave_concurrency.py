"""Diagnostic: how many wavefronts of the solve kernel are alive at the same time."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
if len(sys.argv) > 2:  # a build with -DILQR_QP_PROF: also boxQP fallback counts per wave
    _hip.LIB_PATH = os.path.abspath(sys.argv[2])
dev = torch.device("cuda:0")
Q, R, Qt, xg = bench.cartpole_cost()
M = _hip.Model(100, Q, R, Qt, xg, None, [-5.0], [5.0])
x0_h, us_h = bench.make_batch(B, 0)
x0 = torch.tensor(x0_h, device=dev); us = torch.tensor(us_h, device=dev)
lpw = int(os.environ.get("ILQR_SOLVE_LPW", 32 if B <= 32768 else 64))  # as the library chooses
nw = (B + lpw - 1) // lpw
log = torch.zeros((nw, 4), dtype=torch.int64, device=dev)
O = _hip.Opts()
_hip.solve(M, O, x0, us); torch.cuda.synchronize()
_hip.check(_hip.lib().ilqr_debug_wave_log(C.c_void_p(log.data_ptr())))
qp_wave = hasattr(_hip.lib(), "ilqr_debug_qp_wave")
if qp_wave:
    _hip.lib().ilqr_debug_qp_wave(None, 1)
res = _hip.solve(M, O, x0, us); torch.cuda.synchronize()
if qp_wave:
    qw = (C.c_uint * (2 * 8192))()
    _hip.lib().ilqr_debug_qp_wave(qw, 0)
    qw = np.array(qw[:]).reshape(2, 8192)[:, :nw]
_hip.check(_hip.lib().ilqr_debug_wave_log(None))
L = log.cpu().numpy()
t0 = L[:, 0].min()
start = (L[:, 0] - t0) / 1e5  # ms (100 MHz)
end = (L[:, 1] - t0) / 1e5
print("waves", nw, "kernel span %.1f ms" % end.max(), "wave life: mean %.1f min %.1f max %.1f ms" % ((end - start).mean(), (end - start).min(), (end - start).max()))
print("start times: %.1f%% within first 1 ms; latest start %.1f ms" % (100 * (start < 1).mean(), start.max()))
for t in (1, 50, 100, 150, 200, 300, 400, 500):
    print("  alive at %3d ms: %d" % (t, int(((start <= t) & (end > t)).sum())))
hw = L[:, 2]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
key = L[:, 3] * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
first = start < 1
print("distinct SIMDs used by waves started in the first ms:", len(np.unique(key[first])), "of", int(first.sum()), "waves")
print("distinct CUs:", len(np.unique(key[first] // 10)))
iters = res["iters"].cpu().numpy()[:nw * lpw].reshape(nw, -1).max(axis=1) if B % lpw == 0 else res["iters"].cpu().numpy()
print("per-wave max iterations: min %d mean %.1f" % (iters.min(), iters.mean()))
# wave lifetime by how many waves share the wave's CU / SIMD
cukey = key // 10
life = end - start
for name, k_ in (("CU", cukey), ("SIMD", key)):
    uniq, inv, cnt = np.unique(k_, return_inverse=True, return_counts=True)
    share = cnt[inv]
    print("waves per %s -> count of waves, mean / max life (ms):" % name)
    for s_ in np.unique(share):
        m = share == s_
        print("   %2d: %5d waves  %.1f / %.1f" % (s_, int(m.sum()), life[m].mean(), life[m].max()))
xcc = L[:, 3]
print("waves per XCC:", np.bincount(xcc.astype(int)))
q = np.percentile(life, [5, 25, 50, 75, 95, 100])
print("life percentiles 5/25/50/75/95/100: " + " ".join("%.1f" % v for v in q))

if qp_wave:
    life_ = end - start
    print("boxQP literal-loop fallbacks per wave: mean %.0f max %d; rejected-candidate blocks: mean %.0f max %d"
          % (qw[0].mean(), qw[0].max(), qw[1].mean(), qw[1].max()))
    print("correlation of wave life with fallbacks %.2f, with rejected-candidate blocks %.2f"
          % (np.corrcoef(life_, qw[0])[0, 1], np.corrcoef(life_, qw[1])[0, 1]))
    order = np.argsort(-life_)[:5]
    for w in order:
        print("   wave %4d life %.1f ms: fallbacks %d, rejected blocks %d" % (w, life_[w], qw[0][w], qw[1][w]))
    A_ = np.stack([np.ones(nw), qw[0], qw[1]], 1)
    coef = np.linalg.lstsq(A_, life_, rcond=None)[0]
    print("least squares: life = %.1f ms + %.2f us per fallback + %.2f us per rejected block" % (coef[0], coef[1] * 1e3, coef[2] * 1e3))

print("slowest waves (wave, life ms, xcc, se, sh, cu, simd):")
for w in np.argsort(-life)[:10]:
    print("   %5d %.1f  xcc %d se %d sh %d cu %2d simd %d" % (w, life[w], xcc[w], se[w], sh[w], cu[w], simd[w]))
print("mean life by XCC:", " ".join("%.1f" % life[xcc == i].mean() for i in range(8)))
print("mean life by CU id:", " ".join("%d:%.1f" % (i, life[cu == i].mean()) for i in np.unique(cu)))
