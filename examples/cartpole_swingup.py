"""Cart-pole swing-up with control limits on the GPU path: the single problem of the reference's
examples/cartpole_example.py (same model, cost, horizon and callback), then the same solver on a
batch of random initial states in one launch.

    python examples/cartpole_swingup.py [batch=4096]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr import iLQR  # noqa: E402
from ilqr.cartpole import CartPoleDynamics  # noqa: E402
from ilqr.ilqr_cost import QRCost  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dynamics = CartPoleDynamics(0.05)
    x_goal = dynamics.augment_state(np.array([0.0, 0.0, 0.0, 0.0]))
    Q = 100 * np.eye(dynamics.state_dimension)
    Q[1, 1] = Q[4, 4] = 10.0
    cost = QRCost(Q, np.array([[5.0]]), Q_terminal=100 * np.eye(dynamics.state_dimension), x_goal=x_goal)
    N = 100

    # one problem, iteration by iteration
    x0 = dynamics.augment_state(np.array([0.0, 0.0, np.pi, 0.0]))
    us_init = np.random.RandomState(0).uniform(-1, 1, (N, dynamics.control_dimension))
    J_hist = []

    def on_iteration(iteration_count, xs, us, J_opt, accepted, converged):
        J_hist.append(J_opt)
        info = "converged" if converged else ("accepted" if accepted else "failed")
        if iteration_count % 20 == 0 or converged:
            print("iteration", iteration_count, info, J_opt, dynamics.reduce_state(xs[-1]))

    xs, us = iLQR(dynamics, cost, N).update_u(x0, us_init, on_iteration=on_iteration)
    print("single problem: %d iterations, final cost %.4f, final state %s, max |u| %.3f"
          % (len(J_hist), J_hist[-1], np.round(dynamics.reduce_state(xs[-1]), 4), np.abs(us).max()))

    # a batch of independent problems in one launch
    r = np.random.RandomState(1)
    st = np.stack([r.uniform(-1, 1, batch), r.uniform(-1, 1, batch), r.uniform(-np.pi, np.pi, batch),
                   r.uniform(-1, 1, batch)], axis=1)
    solver = iLQR(dynamics, cost, N)
    xs_b, us_b = solver.update_u(dynamics.augment_state(st), np.random.RandomState(2).uniform(-1, 1, (batch, N, 1)))
    final = dynamics.reduce_state(xs_b[:, -1])
    upright = np.abs(np.remainder(final[:, 2] + np.pi, 2 * np.pi) - np.pi) < 0.1
    print("batch of %d: mean iterations %.1f, converged %.1f %%, pole within 0.1 rad of upright at the end: %.1f %%"
          % (batch, solver.iterations_.mean(), 100.0 * (solver.status_ == 1).mean(), 100.0 * upright.mean()))


if __name__ == "__main__":
    main()
