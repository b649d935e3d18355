"""Cart-pole model (mirror of ilqr/cartpole.py).

State [x, x', sin(theta), cos(theta), theta'], action [F]; theta = 0 points down
(ilqr/cartpole.py:21-25).  The reference builds the model symbolically with Theano;
here the same expression and its Jacobian are device functions of the HIP library,
and f / f_x / f_u below evaluate them on the GPU through the C ABI.
"""
import numpy as np

from .ilqr_dynamics import Dynamics


def _is_torch(a):
    return type(a).__module__.split(".")[0] == "torch"


class CartPoleDynamics(Dynamics):
    """cartpole dynamics (ilqr/cartpole.py:6-95)."""

    device_model = "cartpole"

    def __init__(self, dt, max_control=5, mc=1.0, mp=0.1, l=1.0, g=9.8, **kwargs):
        if kwargs:
            raise TypeError("unsupported arguments %s (the Theano compile options of the reference "
                            "have no meaning here)" % sorted(kwargs))
        self.gravity = g
        self.masscart = mc
        self.masspole = mp
        self.total_mass = self.masspole + self.masscart
        self.length = l  # actually half the pole's length
        self.polemass_length = self.masspole * self.length
        self.upper_bound = np.array([max_control], dtype=np.float64)
        self.lower_bound = np.array([max_control * -1], dtype=np.float64)
        self.tau = dt
        self.state = None

    @property
    def state_dimension(self):
        return 5

    @property
    def control_dimension(self):
        return 1

    @property
    def has_hessians(self):
        return False

    # -- point evaluations (GPU, through the stage kernels) ---------------------
    def _model(self, T=1):
        from . import _hip
        eye = np.eye(5)
        return _hip.Model(T, eye, np.eye(1), eye, None, None, self.lower_bound, self.upper_bound, dt=self.tau,
                          mc=self.masscart, mp=self.masspole, l=self.length, g=self.gravity)

    def _points(self, x, u):
        import torch
        x = np.asarray(x, dtype=np.float64).reshape(-1, 5)
        u = np.asarray(u, dtype=np.float64).reshape(-1, 1)
        dev = torch.device("cuda", torch.cuda.current_device())
        return torch.tensor(x, device=dev), torch.tensor(u, device=dev), x.shape[0]

    def f(self, x, u, i):
        """Next state; x [5] (or [B,5]), u [1] (or [B,1])."""
        from . import _hip
        single = np.asarray(x).ndim == 1
        xt, ut, B = self._points(x, u)
        xs_bm, _ = _hip.rollout_cost_bm(self._model(1), B, _hip.pack_bm(xt), _hip.pack_bm(ut.reshape(B, 1, 1)))
        out = _hip.unpack_bm(xs_bm, B, (2, 5))[:, 1].cpu().numpy()
        return out[0] if single else out

    def _jac(self, x, u):
        from . import _hip
        xt, ut, B = self._points(x, u)
        M = self._model(1)
        xs_bm, _ = _hip.rollout_cost_bm(M, B, _hip.pack_bm(xt), _hip.pack_bm(ut.reshape(B, 1, 1)))
        d = _hip.deriv_bm(M, B, xs_bm, _hip.pack_bm(ut.reshape(B, 1, 1)))
        fx = _hip.unpack_bm(d["fx"], B, (5, 5)).cpu().numpy()
        fu = _hip.unpack_bm(d["fu"], B, (5, 1)).cpu().numpy()
        return fx, fu

    def f_x(self, x, u, i):
        fx, _ = self._jac(x, u)
        return fx[0] if np.asarray(x).ndim == 1 else fx

    def f_u(self, x, u, i):
        _, fu = self._jac(x, u)
        return fu[0] if np.asarray(x).ndim == 1 else fu

    # -- state (de)augmentation, ilqr/cartpole.py:97-139 ---------------------------
    @classmethod
    def augment_state(cls, state):
        """[x, x', theta, theta'] -> [x, x', sin(theta), cos(theta), theta'] (1-D or 2-D)."""
        if _is_torch(state):
            # device tensors stay on the device; sin / cos are the library's own (the functions
            # the solver kernels use), so a closed loop kept on the GPU is reproducible bit for bit
            import torch
            from . import _hip
            theta = state[..., 2].to(torch.float64).contiguous()
            if theta.is_cuda:
                sin_t, cos_t = _hip.detmath(theta, theta)[:2]
            else:
                sin_t, cos_t = torch.sin(theta), torch.cos(theta)
            return torch.stack([state[..., 0], state[..., 1], sin_t, cos_t, state[..., 3]], dim=-1)
        state = np.asarray(state, dtype=np.float64)
        if state.ndim == 1:
            x, x_dot, theta, theta_dot = state
        else:
            x = state[:, 0].reshape(-1, 1)
            x_dot = state[:, 1].reshape(-1, 1)
            theta = state[:, 2].reshape(-1, 1)
            theta_dot = state[:, 3].reshape(-1, 1)
        return np.hstack([x, x_dot, np.sin(theta), np.cos(theta), theta_dot])

    @classmethod
    def reduce_state(cls, state):
        """[x, x', sin(theta), cos(theta), theta'] -> [x, x', theta, theta'] (1-D or 2-D)."""
        if _is_torch(state):
            import torch
            from . import _hip
            s_, c_ = state[..., 2].to(torch.float64).contiguous(), state[..., 3].to(torch.float64).contiguous()
            theta = _hip.detmath(s_, c_)[2] if s_.is_cuda else torch.atan2(s_, c_)
            return torch.stack([state[..., 0], state[..., 1], theta, state[..., 4]], dim=-1)
        state = np.asarray(state, dtype=np.float64)
        if state.ndim == 1:
            x, x_dot, sin_theta, cos_theta, theta_dot = state
        else:
            x = state[:, 0].reshape(-1, 1)
            x_dot = state[:, 1].reshape(-1, 1)
            sin_theta = state[:, 2].reshape(-1, 1)
            cos_theta = state[:, 3].reshape(-1, 1)
            theta_dot = state[:, 4].reshape(-1, 1)
        theta = np.arctan2(sin_theta, cos_theta)
        return np.hstack([x, x_dot, theta, theta_dot])
