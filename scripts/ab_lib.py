"""A/B timing of differently compiled builds of the library on the benchmark batch:
python scripts/ab_lib.py B lib1.so [lib2.so ...]   (each library in a child process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(B, path):
    import time
    import torch
    sys.path.insert(0, ROOT)
    from ilqr_amd import _hip
    _hip.LIB_PATH = os.path.abspath(path)
    import bench
    x0, us0 = bench.make_batch(B, 0)
    Q, R, Qt, xg = bench.cartpole_cost()
    dx, du = torch.tensor(x0, device="cuda:0"), torch.tensor(us0, device="cuda:0")
    O = _hip.Opts(max_iter=200)
    M = _hip.Model(100, Q, R, Qt, xg, None, [-5.0], [5.0])
    r = _hip.solve(M, O, dx, du)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        _hip.solve(M, O, dx, du, workspace=r["workspace"])
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    print("%-40s B=%d  %.2f ms  (J checksum %.6f)" % (os.path.basename(path), B, min(ts) * 1e3, float(r["J"].sum())),
          flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3])
    else:
        for lib in sys.argv[2:]:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", sys.argv[1], lib], check=True, timeout=300)
