"""Timing of the single swing-up problem of the reference's example script on the GPU path."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr import iLQR
from ilqr.cartpole import CartPoleDynamics
from ilqr.ilqr_cost import QRCost
d = CartPoleDynamics(0.05)
Q = 100 * np.eye(5); Q[1, 1] = Q[4, 4] = 10.0
c = QRCost(Q, np.array([[5.0]]), Q_terminal=100 * np.eye(5), x_goal=d.augment_state(np.array([0.0, 0.0, 0.0, 0.0])))
x0 = d.augment_state(np.array([0.0, 0.0, np.pi, 0.0]))
us = np.random.RandomState(0).uniform(-1, 1, (100, 1))
iLQR(d, c, 100).update_u(x0, us)
ts = []
for _ in range(5):
    t = time.perf_counter(); s = iLQR(d, c, 100); s.update_u(x0, us); ts.append(time.perf_counter() - t)
print("single problem, one launch: %.1f ms (%d iterations, %.3f ms per iteration)" % (min(ts) * 1e3, s.iterations_, min(ts) * 1e3 / s.iterations_))
n = [0]
def cb(*a): n[0] += 1
t = time.perf_counter(); iLQR(d, c, 100).update_u(x0, us, on_iteration=cb); dt = time.perf_counter() - t
print("with on_iteration (one launch per iteration): %.1f ms for %d iterations" % (dt * 1e3, n[0]))
