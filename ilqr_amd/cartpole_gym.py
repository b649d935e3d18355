"""The closed-loop plant of the MPC example (mirror of the physics in
ilqr/cartpole_gym.py:24-34, :60-72, :100-103) without the gym / rendering glue.

``CartPoleContinuousEnv`` keeps the reference's parameter names and ``_state_eq``,
``step``, ``reset``; state is the REDUCED state [x, x', theta, theta'].  Everything is
batched: a single state [4] or a batch [B, 4]; the step runs on the GPU.
"""
import numpy as np

from . import _hip
from .cartpole import CartPoleDynamics


def _is_torch(a):
    return type(a).__module__.split(".")[0] == "torch"


class CartPoleContinuousEnv(object):
    def __init__(self):
        self.gravity = 9.8
        self.masscart = 1.0
        self.masspole = 0.1
        self.total_mass = self.masspole + self.masscart
        self.length = 0.5  # actually half the pole's length
        self.polemass_length = self.masspole * self.length
        self.max_force = 20.0
        self.tau = 0.05  # seconds between state updates
        self.dynamics = CartPoleDynamics(dt=self.tau, max_control=self.max_force, mc=self.masscart,
                                         mp=self.masspole, l=self.length, g=self.gravity)
        self.theta_threshold_radians = 12 * 2 * np.pi / 360
        self.x_threshold = 2.4
        self.state = None

    def _state_eq(self, st, u):
        """One explicit-Euler step of the plant (cartpole_gym.py:60-72)."""
        import torch
        as_torch = _is_torch(st)
        single = (st.dim() if as_torch else np.ndim(st)) == 1
        dev = torch.device("cuda", torch.cuda.current_device())
        s = st if as_torch else torch.tensor(np.asarray(st, dtype=np.float64), device=dev)
        a = u if _is_torch(u) else torch.tensor(np.asarray(u, dtype=np.float64), device=dev)
        s = s.to(device=dev, dtype=torch.float64).reshape(-1, 4).contiguous()
        a = a.to(device=dev, dtype=torch.float64).reshape(-1).contiguous()
        out = _hip.plant_step(s, a, gravity=self.gravity, masscart=self.masscart, masspole=self.masspole,
                              length=self.length, tau=self.tau)
        out = out[0] if single else out
        return out if as_torch else out.cpu().numpy()

    def step(self, action):
        self.state = self._state_eq(self.state, action)
        st = self.state
        x, theta = st[..., 0], st[..., 2]
        done = (x < -self.x_threshold) | (x > self.x_threshold) | (theta < -self.theta_threshold_radians) \
            | (theta > self.theta_threshold_radians)
        reward = 1.0 - 1.0 * done
        return self.state, reward, done, {}

    def reset(self, batch=None):
        """[0, 0, pi/2, 0] (cartpole_gym.py:100-103), optionally replicated ``batch`` times."""
        st = np.array([0.0, 0.0, 0.5 * np.pi, 0.0])
        self.state = st if batch is None else np.tile(st, (batch, 1))
        return np.array(self.state)
