"""Times the per-model-library solver (user-defined dynamics) against the built-in cart-pole path
on the benchmark batch: python scripts/user_bench.py [B]."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ilqr_amd import _hip, user_model as um  # noqa: E402
import bench  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    T = 100
    x0, us0 = bench.make_batch(B, 0)
    Q, R, Qt, xg = bench.cartpole_cost()
    dx, du = torch.tensor(x0, device="cuda:0"), torch.tensor(us0, device="cuda:0")
    O = _hip.Opts(max_iter=200)
    Mh = _hip.Model(T, Q, R, Qt, xg, None, [-5.0], [5.0])
    Mu = um.UserModel(um.load("cartpole_user"), T, [0.05, 1.0, 0.1, 1.0, 9.8], Q, R, Qt, xg, None, [-5.0], [5.0])
    a = _hip.solve(Mh, O, dx, du)
    b = um.solve(Mu, O, dx, du)
    same = all(torch.equal(a[k], b[k]) for k in ("xs", "us", "J", "iters", "status"))
    tb = timed(lambda: _hip.solve(Mh, O, dx, du, workspace=a["workspace"]))
    tu = timed(lambda: um.solve(Mu, O, dx, du, workspace=b["workspace"]))
    print("B=%d cart-pole: built-in %.1f ms (%.0f solves/s), as user model %.1f ms (%.0f solves/s), identical=%s"
          % (B, tb * 1e3, B / tb, tu * 1e3, B / tu, same))


if __name__ == "__main__":
    main()
