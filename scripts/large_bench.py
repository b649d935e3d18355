#!/usr/bin/env python3
"""Timing of the large-state path (n = 32, m = 8, T = 200, box limits +-1): whole solves and
the two stage kernels.   python scripts/large_bench.py [B=4096] [T=200]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# fp64 operations of one backward step as the reference computes it (both K^T Q_ux and Q_ux^T K,
# both f_x^T V_xx f_x products): 4 n^3 + 10 m n^2 + 6 m^2 n + m^3 / 3 + O(n^2), n = 32, m = 8
FLOP_BW_STEP = 4 * 32 ** 3 + 10 * 8 * 32 ** 2 + 6 * 8 ** 2 * 32 + 8 ** 3 / 3.0
# one rollout step for one step size: A x, W x (2 n^2 each), B u, K dx (2 n m each), cost 2 n^2 + 2 m^2
FLOP_FW_STEP = 3 * 2 * 32 ** 2 + 2 * 2 * 32 * 8 + 2 * 8 ** 2
# MFMA instructions actually issued (16 x 16 x 4, 2048 flop each) per backward step / per 16-wide rollout step
# (162 = 208 less the matrix-vector products W x and l_x, formed for 16 steps at a time, and f_x^T V_x, which rides as
# a ninth column of the Q_ux product)
MFMA_BW_STEP, MFMA_FW_STEP = 162, 76


def main():
    import torch
    from ilqr_amd import _hip
    from ilqr_amd.synthetic import SyntheticWorkload
    if os.environ.get("ILQR_LIB"):  # A/B of differently compiled builds: ILQR_LIB=build/variant.so
        _hip.LIB_PATH = os.path.abspath(os.environ["ILQR_LIB"])
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda:0")
    wl = SyntheticWorkload(T=T)
    M = wl.device_model()
    x0_h, us_h = wl.batch(B)
    x0, us = torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev)
    O = _hip.Opts()

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, out

    t_roll, (xs, J) = timed(lambda: _hip.large_rollout_cost(M, x0, us))
    mu = torch.ones(B, dtype=torch.float64, device=dev)
    t_bw, bw = timed(lambda: _hip.large_backward(M, O, xs, us, mu, want_masks=False))
    al = 1.1 ** (-np.arange(30.0) ** 2)  # ilqr_core.py:267
    t_fw16, _ = timed(lambda: _hip.large_forward_ls(M, al[:16], xs, us, bw["k"], bw["K"]))
    t_solve, out = timed(lambda: _hip.large_solve(M, O, x0, us), reps=2)
    iters = out["iters"].cpu().numpy().astype(np.int64)
    status = out["status"].cpu().numpy()
    waves = min(B, 1024)
    print("B %d T %d" % (B, T))
    print("rollout + cost      %8.3f ms" % (t_roll * 1e3))
    print("backward sweep      %8.3f ms  -> %.2f TFLOP/s algorithmic (%.0f kflop/step), MFMA issue %.1f %% of the "
          "64-cycle rate on %d SIMDs" % (t_bw * 1e3, B * T * FLOP_BW_STEP / t_bw / 1e12, FLOP_BW_STEP / 1e3,
                                          100.0 * B * T * MFMA_BW_STEP * 64 / (t_bw * 2.4e9 * waves), waves))
    print("16-alpha rollout    %8.3f ms  -> MFMA issue %.1f %%" % (t_fw16 * 1e3, 100.0 * B * T * MFMA_FW_STEP * 64 /
                                                                    (t_fw16 * 2.4e9 * waves)))
    print("solve               %8.3f ms  -> %.0f solves/s, %.1f iterations/solve (max %d), converged %.1f %%, "
          "%.2f ms per iteration of the batch" % (t_solve * 1e3, B / t_solve, iters.mean(), iters.max(),
                                                  100.0 * np.mean(status == 1), t_solve * 1e3 / iters.max()))
    # How far the launch is from a perfectly balanced one: a problem takes iters x (time per iteration); wavefronts
    # take problems from a queue in index order (list scheduling on 1024 wavefronts).
    import heapq

    def makespan(jobs, slots):
        heap = [0.0] * slots
        for j in jobs:
            heapq.heappush(heap, heapq.heappop(heap) + j)
        return max(heap)
    per_it = t_solve / makespan(iters, 1024)
    print("list-scheduling model: %.2f ms per iteration of one problem; if four fixed problems shared a workgroup: %.0f ms; "
          "as launched (queue): %.0f ms; perfectly divisible: %.0f ms; longest problem alone: %.0f ms"
          % (per_it * 1e3, per_it * makespan(iters[:B // 4 * 4].reshape(-1, 4).max(1), 256) * 1e3, t_solve * 1e3,
             per_it * iters.sum() / 1024 * 1e3, per_it * iters.max() * 1e3))


if __name__ == "__main__":
    main()
