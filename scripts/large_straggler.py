#!/usr/bin/env python3
"""Which problems of the large-state batch run longest, and what their iterations look like (boxQP passes and
Armijo trips per backward sweep, step size index of the line search).   python scripts/large_straggler.py [B=8192]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from ilqr_amd import _hip
    from ilqr_amd.synthetic import SyntheticWorkload
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    dev = torch.device("cuda:0")
    wl = SyntheticWorkload(T=200)
    M = wl.device_model()
    x0_h, us_h = wl.batch(B)
    out = _hip.large_solve(M, _hip.Opts(), torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev), trace=True)
    it = out["iters"].cpu().numpy()
    order = np.argsort(-it)
    print("longest problems:", [(int(b), int(it[b])) for b in order[:12]])
    print("histogram of iteration counts (>= 20):", np.bincount(it)[20:])
    tr = {k: v.cpu().numpy() for k, v in out["trace"].items()}
    for b in order[:2]:
        n = it[b]
        print("problem %d: %d iterations" % (b, n))
        print("  accepted      ", tr["accepted"][:n, b].tolist())
        print("  alpha_idx     ", tr["alpha_idx"][:n, b].tolist())
        print("  qp_iter/200   ", np.round(tr["qp_iter"][:n, b] / 200.0, 2).tolist())
        print("  qp_backtr/200 ", np.round(tr["qp_backtrack"][:n, b] / 200.0, 1).tolist())
        print("  n_clamped     ", tr["n_clamped"][:n, b].tolist())
        print("  J             ", tr["J"][:n:8, b].tolist())


if __name__ == "__main__":
    main()
