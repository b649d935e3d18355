#!/usr/bin/env python3
"""Throughput of a user-written 20-state / 5-control model on the large-state path (tests/models/long_chain_user.h,
library built by __graft_entry__.build()):   python scripts/large_user_bench.py [B=4096] [T=50]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from ilqr_amd import _hip, user_model as um
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda:0")
    lib = um.load_large("long_chain_user")
    Q = np.diag([4.0] * 10 + [1.0] * 10)
    R = 0.2 * np.eye(5) + 0.02 * np.ones((5, 5))
    xg = np.concatenate([0.3 * np.arange(1, 11), np.zeros(10)])
    M = um.LargeUserModel(lib, T, [0.05, 4.0, 6.0, 0.3], Q, R, 30.0 * np.eye(20), xg, None, [-1.5] * 5, [1.5] * 5)
    r = np.random.RandomState(7)
    x0 = np.concatenate([np.cumsum(r.uniform(0.0, 0.5, (B, 10)), axis=1), r.uniform(-0.5, 0.5, (B, 10))], axis=1)
    us = r.uniform(-1.0, 1.0, (B, T, 5))
    x0, us = torch.tensor(x0, device=dev), torch.tensor(us, device=dev)
    O = _hip.Opts(max_iter=100)
    out = _hip.large_solve(M, O, x0, us)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = _hip.large_solve(M, O, x0, us, workspace=out["workspace"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it = out["iters"].cpu().numpy()
    print("user model 20 x 5, T = %d, %d problems: %.1f ms, %.0f solves/s, %.1f iterations per solve (max %d), "
          "converged %.1f %%" % (T, B, dt * 1e3, B / dt, it.mean(), it.max(),
                                 100.0 * (out["status"].cpu().numpy() == 1).mean()))


if __name__ == "__main__":
    main()
