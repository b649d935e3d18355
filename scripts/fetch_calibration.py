"""One k_solve launch with a KNOWN byte count, for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on the solve
kernel's own access shape (8 bytes per lane, 256- or 512-byte wave accesses into the wave-tiled workspace):

    python scripts/fetch_calibration.py <problems>       (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)

With max_iter = 1 every problem makes exactly one iteration, so the launch reads and writes exactly
    initial rollout: reads x0, us                          writes xs[1..T]
    backward sweep : reads xs, us                          writes k, K
    alpha_0 rollout: reads xs, us, k, K                    writes xs, us
per problem (a rejected first iteration adds one cooperative line-search pass: reads only, mostly cache hits).
Prints the byte counts as JSON."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ilqr_amd import _hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = bench.T
dev = torch.device("cuda:0")
Q, R, Qt, xg = bench.cartpole_cost()
M = _hip.Model(T, Q, R, Qt, xg, None, [-5.0], [5.0])
x0_h, us_h = bench.make_batch(B, 0)
res = _hip.solve(M, _hip.Opts(max_iter=1), torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev), trace=True)
torch.cuda.synchronize()
acc = res["trace"]["accepted"][0].cpu().numpy()
reads = 8 * ((5 + T) + ((T + 1) * 5 + T) + ((T + 1) * 5 + T + T * 6))
writes = 8 * (T * 5 + T * 6 + ((T + 1) * 5 + T))
print(json.dumps({"batch": B, "read_bytes": reads * B, "write_bytes": writes * B,
                  "first_iteration_accepted_fraction": float((acc == 1).mean())}))
