#!/usr/bin/env python3
"""Timing of the LDS-resident wide path (ilqr_xl_*: one problem per workgroup) for a model size, and -- where both
apply (n <= 32, m <= 8) -- of the register-resident path (one problem per wavefront) on the same batch.
   python scripts/xl_bench.py [n=64] [m=16] [B=1024] [T=200] [max_iter=200]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from ilqr_amd import _hip
    from ilqr_amd.synthetic import SyntheticWorkload
    if os.environ.get("ILQR_LIB"):  # A/B of differently compiled builds: ILQR_LIB=build/variant.so
        _hip.LIB_PATH = os.path.abspath(os.environ["ILQR_LIB"])
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    n_it = int(sys.argv[5]) if len(sys.argv) > 5 else 200
    dev = torch.device("cuda:0")
    wl = SyntheticWorkload(n=n, m=m, T=T)
    x0_h, us_h = wl.batch(B)
    x0, us = torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev)
    O = _hip.Opts(max_iter=n_it)
    al = 1.1 ** (-np.arange(30.0) ** 2)

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, out

    flop_bw = 4 * n ** 3 + 10 * m * n ** 2 + 6 * m ** 2 * n + m ** 3 / 3.0
    for wide in ((True, False) if (n <= 32 and m <= 8) else (True,)):
        M = _hip.LargeModel(T, wl.A, wl.B, wl.W, wl.Q, wl.R, wl.Qt, wl.x_goal, wl.u_goal, wl.lower, wl.upper,
                            dt=wl.dt, sig_c=wl.sig_c, wide=wide)
        t_roll, (xs, J) = timed(lambda: _hip.large_rollout_cost(M, x0, us))
        mu = torch.ones(B, dtype=torch.float64, device=dev)
        t_bw, bw = timed(lambda: _hip.large_backward(M, O, xs, us, mu, want_masks=False))
        t_fw, _ = timed(lambda: _hip.large_forward_ls(M, al[:4], xs, us, bw["k"], bw["K"]))
        t_solve, out = timed(lambda: _hip.large_solve(M, O, x0, us), reps=1)
        it = out["iters"].cpu().numpy().astype(np.int64)
        slots = min(B, 256 if wide else 1024)
        rounds = -(-B // slots)
        print("%s path, %d x %d, B %d, T %d" % ("wide (workgroup per problem)" if wide else
                                                  "register (wavefront per problem)", n, m, B, T))
        print("  rollout + cost   %8.3f ms" % (t_roll * 1e3))
        print("  backward sweep   %8.3f ms = %.0f cycles per step and problem slot (%d rounds), %.2f TFLOP/s algorithmic"
              % (t_bw * 1e3, t_bw * 2.4e9 / T / rounds, rounds, B * T * flop_bw / t_bw / 1e12))
        print("  4-alpha rollout  %8.3f ms = %.0f cycles per step and slot" % (t_fw * 1e3, t_fw * 2.4e9 / T / rounds))
        print("  solve            %8.3f ms -> %.0f solves/s; iterations mean %.1f max %d; %.2f TFLOP/s algorithmic "
              "(backward only)" % (t_solve * 1e3, B / t_solve, it.mean(), it.max(),
                                   it.sum() * T * flop_bw / t_solve / 1e12))


if __name__ == "__main__":
    main()
