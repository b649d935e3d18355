"""Receding-horizon control of the cart-pole plant on the GPU path, after the reference's
examples/cartpole_example_mpc.py (same plant, model, cost, horizon N = 15, first solve 200 iterations,
then 10 per control step, the measured state fed back with set_state), for a batch of closed loops
at once; all tensors stay on the device.

    python examples/cartpole_mpc.py [loops=256] [steps=100]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr import RecedingHorizonController, iLQR  # noqa: E402
from ilqr.cartpole_gym import CartPoleContinuousEnv  # noqa: E402
from ilqr.ilqr_cost import QRCost  # noqa: E402


def main():
    loops = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    env = CartPoleContinuousEnv()
    dynamics = env.dynamics  # CartPoleDynamics(dt=0.05, max_control=20, mc=1, mp=0.1, l=0.5, g=9.8)
    Q = 100.0 * np.eye(5)
    Q[0, 0] = Q[1, 1] = 10.0
    Q[4, 4] = 80.0
    cost = QRCost(Q, np.array([[1.0]]), Q_terminal=100.0 * np.eye(5),
                  x_goal=dynamics.augment_state(np.array([0.0, 0.0, 0.0, 0.0])))
    N = 15
    r = np.random.RandomState(3)
    dev = torch.device("cuda", torch.cuda.current_device())
    state = torch.tensor(env.reset(batch=loops) + r.uniform(-0.05, 0.05, (loops, 4)), device=dev)
    env.state = state
    us_init = torch.tensor(r.uniform(-1.0, 1.0, (loops, N, 1)), device=dev)
    solver = iLQR(dynamics, cost, N)
    controller = RecedingHorizonController(dynamics.augment_state(state), us_init, solver, env)
    for i, (xs, us) in enumerate(controller.control(step_size=1, initial_n_iterations=200,
                                                    subsequent_n_iterations=10)):
        state, _, _, _ = env.step(us[:, 0])
        controller.set_state(dynamics.augment_state(state))
        if i % 20 == 0 or i + 1 == steps:
            s = state.cpu().numpy()
            print("step %3d: mean |x| %.3f m, mean pole angle %.3f rad, mean iterations %.1f, control of loop 0: %+.3f"
                  % (i, np.abs(s[:, 0]).mean(), np.abs(s[:, 2]).mean(), float(np.mean(solver.iterations_.cpu().numpy()
                     if hasattr(solver.iterations_, "cpu") else solver.iterations_)), float(us[0, 0, 0])))
        if i + 1 >= steps:
            break


if __name__ == "__main__":
    main()
