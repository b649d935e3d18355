"""Closed-loop MPC with the whole control step captured in a HIP graph.

One control step of the receding-horizon loop of the reference's MPC example
(examples/cartpole_example_mpc.py:66-79 with ilqr_core.py:422-466) is ~30 small launches when
driven through the generator: augment the measured state, pack, solve (<= 10 warm-started
iterations), unpack, apply the first control to the plant, shift the control sequence.  At a few
thousand closed loops the launches, not the kernels, set the step time.  Here the step is four
launches (pack, solve, unpack, and one kernel for plant + augment + shift), captured once in a
graph and replayed.  Results are those of ``RecedingHorizonController`` + ``set_state`` bit
for bit (same kernels, same order).
"""
import numpy as np

from . import _hip


class GraphedClosedLoop(object):
    def __init__(self, controller, env, state, us_init, subsequent_n_iterations=10, tolerance=1e-4,
                 line_search_steps=30, scale=0.1):
        """controller: an ``iLQR``; env: ``CartPoleContinuousEnv``; state [B, 4] and us_init [B, N, 1]
        device tensors.  Call ``first(n_iterations)`` once, then ``step()`` per control step."""
        import torch
        self.c, self.env = controller, env
        self.dyn = controller.dynamics
        self.B, self.N = state.shape[0], controller.N
        dev = state.device
        self.st = state.to(torch.float64).contiguous().clone()
        self.us = us_init.to(torch.float64).reshape(self.B, self.N, 1).contiguous().clone()
        self.M = controller._model()
        self.kw = dict(tol=tolerance, line_search_steps=line_search_steps, scale=scale, mu_min=controller.mu_min_,
                       mu_max=controller.mu_max_ if controller.mu_max_ else 0.0, delta0=controller.delta_0_)
        self.O = _hip.Opts(max_iter=subsequent_n_iterations, **self.kw)
        n = 5
        self.delta = torch.full((self.B,), float(controller.delta_0_), dtype=torch.float64, device=dev)
        self.k = torch.zeros((self.B, self.N, 1), dtype=torch.float64, device=dev)
        self.K = torch.zeros((self.B, self.N, 1, n), dtype=torch.float64, device=dev)
        self.out = dict(xs=torch.empty((self.B, self.N + 1, n), dtype=torch.float64, device=dev),
                        us=torch.empty((self.B, self.N, 1), dtype=torch.float64, device=dev),
                        J=torch.empty(self.B, dtype=torch.float64, device=dev),
                        mu=torch.empty(self.B, dtype=torch.float64, device=dev),
                        iters=torch.empty(self.B, dtype=torch.int32, device=dev),
                        status=torch.empty(self.B, dtype=torch.int32, device=dev))
        self.ws = torch.empty(_hip.solve_workspace_bytes(self.M, self.B), dtype=torch.uint8, device=dev)
        self.u_applied = torch.empty(self.B, dtype=torch.float64, device=dev)
        self.x0 = self.dyn.augment_state(self.st).contiguous()  # augmented state the next solve starts from
        self.graph = None

    def _one_step(self, O):
        _hip.solve(self.M, O, self.x0, self.us, delta=self.delta, k=self.k, K=self.K, workspace=self.ws, out=self.out)
        # plant step with the first control, augmented state and control seed of the next solve
        # (ilqr_core.py:457-465, cartpole_gym.py:60-72, cartpole.py:97-116): one launch
        _hip.mpc_advance(self.st, self.out["us"], self.us, self.u_applied, self.x0, gravity=self.env.gravity,
                         masscart=self.env.masscart, masspole=self.env.masspole, length=self.env.length,
                         tau=self.env.tau)

    def first(self, initial_n_iterations=200):
        """The first solve (more iterations), eagerly; then capture the subsequent step."""
        import torch
        self._one_step(_hip.Opts(max_iter=initial_n_iterations, **self.kw))
        torch.cuda.synchronize()
        # warm the allocator on a side stream, then capture (the usual torch recipe)
        keep = [t.clone() for t in (self.st, self.us, self.delta, self.k, self.K, self.u_applied, self.x0)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._one_step(self.O)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._one_step(self.O)
        torch.cuda.synchronize()
        for t, v in zip((self.st, self.us, self.delta, self.k, self.K, self.u_applied, self.x0), keep):
            t.copy_(v)  # capture and warm-up must not advance the loop
        return self.st, self.u_applied

    def step(self):
        """One control step: returns (state [B, 4], applied control [B]) views, valid until the next call."""
        self.graph.replay()
        return self.st, self.u_applied
