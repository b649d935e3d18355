#!/usr/bin/env python3
"""Receding-horizon cart-pole MPC, many closed loops at once, everything resident on the GPU
(examples/cartpole_example_mpc.py:66-102 of the reference, batched).

  python scripts/mpc_closed_loop.py [B=4096] [steps=200]

Each control step: one batched update_u (200 iterations for the first, 10 afterwards, warm-started
from the shifted previous solution and the carried delta_/k_/K_), the first control applied to the
plant (explicit Euler, cartpole_gym.py:60-72), the measured state fed back with set_state.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_problem(N=15):
    from ilqr_amd.cartpole_gym import CartPoleContinuousEnv
    from ilqr_amd.ilqr_cost import QRCost
    env = CartPoleContinuousEnv()
    dyn = env.dynamics
    x_goal = dyn.augment_state(np.array([0.0, 0.0, 0.0, 0.0]))
    Q = 100.0 * np.eye(5)                       # cartpole_example_mpc.py:84-95
    Q[0, 0] = Q[1, 1] = 10.0
    Q[4, 4] = 80.0
    cost = QRCost(Q, np.array([[1.0]]), Q_terminal=100.0 * np.eye(5), x_goal=x_goal)
    return env, dyn, cost


def initial_conditions(B, N, seed=3):
    r = np.random.RandomState(seed)
    st = np.tile(np.array([0.0, 0.0, 0.5 * np.pi, 0.0]), (B, 1)) + r.uniform(-0.05, 0.05, (B, 4))
    us = r.uniform(-1.0, 1.0, (B, N, 1))
    return st, us


def run(B=4096, steps=200, N=15, initial_n_iterations=200, subsequent_n_iterations=10, record=False):
    """Returns dict(states [steps+1, B, 4], controls [steps, B], seconds, iterations [steps, B])."""
    import torch
    from ilqr_amd import RecedingHorizonController, iLQR
    env, dyn, cost = make_problem(N)
    st_h, us_h = initial_conditions(B, N)
    dev = torch.device("cuda", torch.cuda.current_device())
    st = torch.tensor(st_h, device=dev)
    env.state = st
    solver = iLQR(dyn, cost, N)
    ctrl = RecedingHorizonController(dyn.augment_state(st), torch.tensor(us_h, device=dev), solver, env)
    gen = ctrl.control(step_size=1, initial_n_iterations=initial_n_iterations,
                       subsequent_n_iterations=subsequent_n_iterations)
    states, controls, its = [st], [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        xs, us = next(gen)
        st, _, _, _ = env.step(us[:, 0])
        ctrl.set_state(dyn.augment_state(st))
        if record:
            states.append(st)
            controls.append(us[:, 0, 0])
            its.append(solver.iterations_)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = dict(seconds=dt, final_state=st.cpu().numpy())
    if record:
        out.update(states=torch.stack(states).cpu().numpy(), controls=torch.stack(controls).cpu().numpy(),
                   iterations=torch.stack(its).cpu().numpy())
    return out


def run_graphed(B=4096, steps=200, N=15, initial_n_iterations=200, subsequent_n_iterations=10, record=False):
    """Same loop with the control step replayed from a HIP graph (ilqr_amd.mpc_graph)."""
    import torch
    from ilqr_amd import iLQR
    from ilqr_amd.mpc_graph import GraphedClosedLoop
    env, dyn, cost = make_problem(N)
    st_h, us_h = initial_conditions(B, N)
    dev = torch.device("cuda", torch.cuda.current_device())
    loop = GraphedClosedLoop(iLQR(dyn, cost, N), env, torch.tensor(st_h, device=dev), torch.tensor(us_h, device=dev),
                             subsequent_n_iterations=subsequent_n_iterations)
    states, controls = [torch.tensor(st_h, device=dev)], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st, u = loop.first(initial_n_iterations)
    if record:
        states.append(st.clone())
        controls.append(u.clone())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps - 1):
        st, u = loop.step()
        if record:
            states.append(st.clone())
            controls.append(u.clone())
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out = dict(seconds=t2 - t0, seconds_first=t1 - t0, seconds_replayed=t2 - t1, final_state=st.cpu().numpy())
    if record:
        out.update(states=torch.stack(states).cpu().numpy(), controls=torch.stack(controls).cpu().numpy())
    return out


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    run(B=min(B, 256), steps=3)  # warm up
    res = run(B=B, steps=steps, record=True)
    th = res["final_state"][:, 2]
    print("closed loops %d, control steps %d: %.3f s  ->  %.0f closed-loop control steps/s (%.2f ms per step of the batch)"
          % (B, steps, res["seconds"], B * steps / res["seconds"], 1e3 * res["seconds"] / steps))
    g = run_graphed(B=B, steps=steps)
    print("with the control step replayed from a HIP graph: %.3f ms per step of the batch -> %.0f control steps/s; "
          "same final states: %s" % (1e3 * g["seconds_replayed"] / (steps - 1), B * (steps - 1) / g["seconds_replayed"],
                                     bool(np.array_equal(g["final_state"], res["final_state"]))))
    print("mean iterations: first solve %.1f, later %.2f; final theta in [%.2f, %.2f] rad, |x| <= %.2f m"
          % (res["iterations"][0].mean(), res["iterations"][1:].mean(), th.min(), th.max(),
             np.abs(res["final_state"][:, 0]).max()))
