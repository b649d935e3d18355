"""Large-state synthetic model of the solve path: 32 states, 8 controls,

    x+ = x + dt (A x + B u + c phi(W x)),     phi(z) = z / sqrt(1 + z^2)   (element-wise),

with box limits |u| <= max_control.  The reference ships no model of this size; this class plays
the role ``CartPoleDynamics`` (ilqr/cartpole.py) plays for the cart-pole: a ``Dynamics`` whose
f / f_x / f_u exist as device code (``device_model = "sigmoid32"``), so that ``iLQR`` can solve it
on the GPU (one problem per wavefront, fp64 MFMA).  f, f_x, f_u below are host-side numpy
evaluations of the same expressions for inspection; the solver never calls them.
"""
import numpy as np

from .ilqr_dynamics import Dynamics


class SigmoidLinearDynamics(Dynamics):
    device_model = "sigmoid32"

    def __init__(self, A, B, W, dt=0.05, sig_c=2.0, max_control=1.0):
        self.A = np.array(A, dtype=np.float64)
        self.B = np.array(B, dtype=np.float64)
        self.W = np.array(W, dtype=np.float64)
        n, m = self.A.shape[0], self.B.shape[1]
        assert self.A.shape == (n, n) and self.W.shape == (n, n) and self.B.shape == (n, m), "A, W [n, n], B [n, m]"
        assert 1 <= n <= 64 and 1 <= m <= 16, "the device model holds up to 64 states and 16 controls"
        self.tau = float(dt)
        self.sig_c = float(sig_c)
        self.upper_bound = np.full(m, float(max_control))
        self.lower_bound = np.full(m, -float(max_control))

    @property
    def state_dimension(self):
        return self.A.shape[0]

    @property
    def control_dimension(self):
        return self.B.shape[1]

    @property
    def has_hessians(self):
        return False

    def f(self, x, u, i):
        z = self.W.dot(x)
        return x + self.tau * (self.A.dot(x) + self.B.dot(u) + self.sig_c * z / np.sqrt(1.0 + z * z))

    def f_x(self, x, u, i):
        z = self.W.dot(x)
        s = 1.0 + z * z
        g = self.sig_c / (s * np.sqrt(s))
        return np.eye(self.A.shape[0]) + self.tau * (self.A + g[:, None] * self.W)

    def f_u(self, x, u, i):
        return self.tau * self.B


class SyntheticWorkload(object):
    """The large-state benchmark workload (the reference ships no model of this size): a stable
    random A, random B and W (``RandomState(seed)``), QRCost towards the origin (Q = I, R = 0.1 I,
    Q_terminal = 10 I), box limits |u| <= u_max."""

    def __init__(self, n=32, m=8, T=200, seed=4, dt=0.05, sig_c=2.0, u_max=1.0):
        r = np.random.RandomState(seed)
        G = r.randn(n, n)
        self.n, self.m, self.T, self.dt, self.sig_c = n, m, T, dt, sig_c
        self.A = 0.8 * G / np.abs(np.linalg.eigvals(G)).max() - np.eye(n)
        self.B = r.randn(n, m) / np.sqrt(m)
        self.W = r.randn(n, n) / np.sqrt(n)
        self.Q, self.R, self.Qt = np.eye(n), 0.1 * np.eye(m), 10.0 * np.eye(n)
        self.x_goal, self.u_goal = np.zeros(n), np.zeros(m)
        self.lower, self.upper = np.full(m, -u_max), np.full(m, u_max)

    def device_model(self, device=None):
        from . import _hip
        return _hip.LargeModel(self.T, self.A, self.B, self.W, self.Q, self.R, self.Qt, self.x_goal, self.u_goal,
                               self.lower, self.upper, dt=self.dt, sig_c=self.sig_c, device=device)

    def batch(self, B, seed_x=5, seed_u=6):
        """Initial states ~ U(-1, 1), initial controls ~ U(-0.5, 0.5)."""
        x0 = np.random.RandomState(seed_x).uniform(-1.0, 1.0, (B, self.n))
        us = np.random.RandomState(seed_u).uniform(-0.5, 0.5, (B, self.T, self.m))
        return x0, us
