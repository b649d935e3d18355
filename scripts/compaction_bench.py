"""Chunked solve with compaction of the running problems against the single launch, on the
benchmark workload:  python scripts/compaction_bench.py [B=131072] [chunk ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip, compaction  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
chunks = [int(c) for c in sys.argv[2:]] or [8, 16, 32]
dev = torch.device("cuda:0")
Q, R, Qt, xg = bench.cartpole_cost()
M = _hip.Model(100, Q, R, Qt, xg, None, [-5.0], [5.0])
x0_h, us_h = bench.make_batch(B, 0)
x0, us = torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev)


def timed(fn, reps=2):
    out = fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, out


t1, a = timed(lambda: _hip.solve(M, _hip.Opts(max_iter=200), x0, us))
print("B=%d single launch: %.1f ms, %.0f solves/s" % (B, t1 * 1e3, B / t1))
for c in chunks:
    tc, b = timed(lambda: _hip.solve_chunked(M, _hip.Opts(max_iter=200), x0, us, c))
    same = all(torch.equal(a[k], b[k]) for k in ("xs", "us", "J", "mu", "delta", "iters", "status", "k", "K"))
    print("   C entry point, chunks of %3d: %.1f ms, %.0f solves/s, identical results: %s" % (c, tc * 1e3, B / tc, same))
    tc, b = timed(lambda: compaction.solve_compacting(_hip.solve, M, {}, 200, x0, us, chunk=c))
    same = all(torch.equal(a[k], b[k]) for k in ("xs", "us", "J", "mu", "delta", "iters", "status", "k", "K"))
    print("   chunks of %3d iterations: %.1f ms, %.0f solves/s, identical results: %s" % (c, tc * 1e3, B / tc, same))

# Where the chunked solve's time goes: problems still running at each chunk boundary, and the time of a
# 32-iteration launch on that many problems (fresh problems: an upper bound, nothing finishes inside it).
it = a["iters"].cpu()
c = compaction.COMPACT_CHUNK
total = 0.0
print("chunk  running  launch of %d iterations on that many problems" % c)
for k in range((200 + c - 1) // c):
    n = int((it > k * c).sum())
    if n == 0:
        break
    n_it = min(c, 200 - k * c)
    tk, _ = timed(lambda: _hip.solve(M, _hip.Opts(max_iter=n_it), x0[:n], us[:n]), reps=1)
    total += tk
    print("  %2d  %7d   %.1f ms  (%.3f ms per iteration)" % (k, n, tk * 1e3, tk * 1e3 / n_it))
print("sum %.1f ms; problem-iterations %d -> at the small-batch rate (0.33 ms per iteration of 65536 lanes): %.1f ms"
      % (total * 1e3, int(it.sum()), float(it.sum()) / 65536 * 0.33))
