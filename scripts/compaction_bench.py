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
