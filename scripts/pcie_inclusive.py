#!/usr/bin/env python3
"""Solve rate when the caller hands over HOST buffers (numpy in, numpy out): the benchmark's
workload through iLQR.update_u, host-to-device and device-to-host copies inside the timed span."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ilqr_amd import iLQR  # noqa: E402
from ilqr_amd.cartpole import CartPoleDynamics  # noqa: E402
from ilqr_amd.ilqr_cost import QRCost  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
Q, R, Qt, xg = bench.cartpole_cost()
x0, us = bench.make_batch(B, 0)
dyn = CartPoleDynamics(0.05)
cost = QRCost(Q, R, Q_terminal=Qt, x_goal=xg)
iLQR(dyn, cost, 100).update_u(x0[:256], us[:256])  # warm up
ts = []
for _ in range(3):
    s = iLQR(dyn, cost, 100)
    t0 = time.perf_counter()
    xs, us_out = s.update_u(x0, us)
    ts.append(time.perf_counter() - t0)
t = min(ts)
print("host buffers in and out, %d problems: %.1f ms per batch -> %.0f solves/s (%.1f MB in, %.1f MB out)"
      % (B, t * 1e3, B / t, (x0.nbytes + us.nbytes) / 1e6, (xs.nbytes + us_out.nbytes) / 1e6))
