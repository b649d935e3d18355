"""iLQR solver and receding-horizon driver (mirror of ilqr/ilqr_core.py) on MI355X.

Same constructor, methods, attributes and semantics as the reference classes, with
one extension: every entry point also takes a BATCH of independent problems
(``x0 [B, n]``, ``u_seq_init [B, N, m]``) and solves them in one device launch.
numpy in -> numpy out; torch (ROCm) tensors in -> torch tensors out, no host copy.

The whole ``update_u`` loop runs on the GPU (see ilqr_amd/csrc).  There is no CPU
fallback: without a GPU or without the built library these classes raise.
"""
import abc
import warnings

import numpy as np

from . import _hip
from .ilqr_cost import PathQRCost, QRCost


class BaseTrajOptimizer(abc.ABC):
    """Base trajectory optimizer (ilqr_core.py:12-28)."""

    @abc.abstractmethod
    def update_u(self, x0, u_seq_init, *args, **kwargs):
        """x0 [n], u_seq_init [N, m] -> (xs [N+1, n], us [N, m])."""
        raise NotImplementedError


def _is_torch(a):
    return type(a).__module__.split(".")[0] == "torch"


def _to_numpy(t):
    """Device tensor -> numpy array through page-locked memory (one copy at PCIe speed; large results
    through pageable memory run at a fraction of it).  The array owns its page-locked block, which
    returns to PyTorch's host cache when the array is dropped."""
    import torch
    if t.numel() * t.element_size() < (1 << 20):
        return t.cpu().numpy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t)
    return h.numpy()


class iLQR(BaseTrajOptimizer):
    """Finite Horizon Iterative Linear Quadratic Regulator (ilqr_core.py:30-399)."""

    def __init__(self, sys_dynamics, cost_func, time_steps=15, max_reg=1e10, control_limit=True,
                 use_second_order=False):
        self.dynamics = sys_dynamics
        self.cost = cost_func
        self.N = int(time_steps)
        self.control_limit = control_limit

        if getattr(sys_dynamics, "device_model", None) not in ("cartpole", "sigmoid32", "user", "large_user"):
            raise NotImplementedError(
                "iLQR on the GPU needs a dynamics model with a device implementation "
                "(ilqr_amd.cartpole.CartPoleDynamics, ilqr_amd.synthetic.SigmoidLinearDynamics, or a C model "
                "source compiled with ilqr_amd.user_model); %r evaluates through Python callbacks, which cannot "
                "run inside a kernel" % type(sys_dynamics).__name__)
        if not isinstance(cost_func, (QRCost, PathQRCost)):
            raise NotImplementedError("only QRCost and PathQRCost have a device implementation")
        if isinstance(cost_func, PathQRCost) and cost_func.x_path.shape[-2] != int(time_steps) + 1:
            raise ValueError("x_path holds %d states, the horizon needs %d" % (cost_func.x_path.shape[-2],
                                                                             int(time_steps) + 1))
        if control_limit is not True:
            # In the reference this branch keeps the positive-sign warm-start gains, predicts a
            # cost INCREASE and rejects every iteration (ilqr_core.py:145,162,305-308).
            raise NotImplementedError("control_limit=False is not usable in the reference solver either "
                                      "(wrong-sign gains); only the control-limited path is implemented")
        if use_second_order:
            raise NotImplementedError("second-order (DDP) terms are unreachable in the reference "
                                      "(ilqr_core.py:52) and are not implemented")
        self.use_second_order_ = False
        self.lower_bound = self.dynamics.lower_bound
        self.upper_bound = self.dynamics.upper_bound

        # regularisation schedule constants (ilqr_core.py:58-62)
        self.mu_min_ = 1e-6
        self.mu_max_ = max_reg
        self.mu_ = 1.0
        self.delta_0_ = 2.0
        self.delta_ = self.delta_0_

        n, m = self.dynamics.state_dimension, self.dynamics.control_dimension
        self._gains = {"k": np.zeros((self.N, m)), "K": np.zeros((self.N, m, n))}
        self._gains_np = {}        # numpy views of device-resident gains, made on first access
        self._gains_as_numpy = True
        # per-problem results of the last call (device-side bookkeeping the reference does not have)
        self.iterations_ = None
        self.status_ = None
        self.J_ = None
        self._batch = None
        super().__init__()

    # ------------------------------------------------------------------ helpers
    def _model(self):
        d, c = self.dynamics, self.cost
        if d.device_model == "user":
            from . import user_model
            if isinstance(c, PathQRCost):
                x_path, u_path = c.device_paths(self._device())
                return user_model.UserModel(d.lib, self.N, d.params, c.Q, c.R, c.Q_terminal, None, None,
                                            self.lower_bound, self.upper_bound, x_path=x_path, u_path=u_path)
            return user_model.UserModel(d.lib, self.N, d.params, c.Q, c.R, c.Q_terminal, c.x_goal, c.u_goal,
                                        self.lower_bound, self.upper_bound)
        if d.device_model == "large_user":
            from . import user_model
            if isinstance(c, PathQRCost):
                return user_model.LargeUserModel(d.lib, self.N, d.params, c.Q, c.R, c.Q_terminal, None, None,
                                                 self.lower_bound, self.upper_bound, x_path=c.x_path, u_path=c.u_path,
                                                 device=self._device())
            return user_model.LargeUserModel(d.lib, self.N, d.params, c.Q, c.R, c.Q_terminal, c.x_goal, c.u_goal,
                                             self.lower_bound, self.upper_bound, device=self._device())
        if d.device_model == "sigmoid32":
            # the device copy of the model (ten uploads) is kept while nothing it was built from has changed
            path = isinstance(c, PathQRCost)
            goals = (c.x_path, c.u_path) if path else (c.x_goal, c.u_goal)
            sig = (self.N, float(d.tau), float(d.sig_c), str(self._device()), path) + tuple(
                np.ascontiguousarray(a, dtype=np.float64).tobytes()
                for a in (d.A, d.B, d.W, c.Q, c.R, c.Q_terminal, self.lower_bound, self.upper_bound) + goals)
            cached = getattr(self, "_large_model_cache", None)
            if cached is not None and cached[0] == sig:
                return cached[1]
            if path:
                M = _hip.LargeModel(self.N, d.A, d.B, d.W, c.Q, c.R, c.Q_terminal, None, None, self.lower_bound,
                                    self.upper_bound, dt=d.tau, sig_c=d.sig_c, device=self._device(),
                                    x_path=c.x_path, u_path=c.u_path)
            else:
                M = _hip.LargeModel(self.N, d.A, d.B, d.W, c.Q, c.R, c.Q_terminal, c.x_goal, c.u_goal,
                                    self.lower_bound, self.upper_bound, dt=d.tau, sig_c=d.sig_c, device=self._device())
            self._large_model_cache = (sig, M)
            return M
        kw = dict(dt=d.tau, mc=d.masscart, mp=d.masspole, l=d.length, g=d.gravity)
        if isinstance(c, PathQRCost):
            x_path, u_path = c.device_paths(self._device())
            return _hip.Model(self.N, c.Q, c.R, c.Q_terminal, None, None, self.lower_bound, self.upper_bound,
                              x_path=x_path, u_path=u_path, **kw)
        return _hip.Model(self.N, c.Q, c.R, c.Q_terminal, c.x_goal, c.u_goal, self.lower_bound, self.upper_bound, **kw)

    # k_, K_ (ilqr_core.py:64-65, :394-395): feed-forward and feedback gains of the last solve
    def _gain(self, name):
        v = self._gains[name]
        if _is_torch(v) and self._gains_as_numpy:
            if name not in self._gains_np:
                self._gains_np[name] = _to_numpy(v)
            return self._gains_np[name]
        return v

    def _set_gain(self, name, value):
        self._gains[name] = value
        self._gains_np.pop(name, None)

    k_ = property(lambda self: self._gain("k"), lambda self, v: self._set_gain("k", v))
    K_ = property(lambda self: self._gain("K"), lambda self, v: self._set_gain("K", v))

    def _device(self):
        import torch
        if not torch.cuda.is_available():
            raise _hip.IlqrHipError("no GPU visible: the solve path has no CPU fallback")
        return torch.device("cuda", torch.cuda.current_device())

    def _to_dev(self, a, shape):
        import torch
        if _is_torch(a):
            return a.to(device=self._device(), dtype=torch.float64).reshape(shape).contiguous()
        return torch.tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape)),
                            device=self._device())

    def _state_for_batch(self, B):
        """Solver state carried across calls (delta_, k_, K_): per-problem copies on the device.
        State left by a single-problem call (or a fresh solver) is broadcast to the batch."""
        import torch
        n, m = self.dynamics.state_dimension, self.dynamics.control_dimension
        dev = self._device()

        def bc(val, shape):
            t = val if _is_torch(val) else torch.tensor(np.asarray(val, dtype=np.float64))
            t = t.to(device=dev, dtype=torch.float64)
            if t.dim() == len(shape):
                t = t.expand((B,) + tuple(shape))
            elif t.shape[0] != B:
                raise ValueError("solver state holds %d problems, this call has %d (use a new iLQR object, "
                                 "or keep the batch size)" % (t.shape[0], B))
            return t.contiguous().clone()
        return bc(self.delta_, ()), bc(self._gains["k"], (self.N, m)), bc(self._gains["K"], (self.N, m, n))

    def _solve(self, M, O, *args, **kwargs):
        if isinstance(M, _hip.LargeModel):
            return _hip.large_solve(M, O, *args, **kwargs)
        if not isinstance(M, _hip.Model):
            from . import user_model
            return user_model.solve(M, O, *args, **kwargs)
        return _hip.solve(M, O, *args, **kwargs)

    # ----------------------------------------------------------------- update_u
    def update_u(self, x0, u_seq_init, n_interations=200, tolerance=1e-4, on_iteration=None,
                 line_search_steps=30, scale=0.1, n_iterations=None):
        """Update the optimal control sequence (ilqr_core.py:241-399).

        x0 [n] or [B, n]; u_seq_init [N, m] or [B, N, m].  ``n_interations`` is the
        reference's (misspelt) keyword; ``n_iterations``, the spelling its own MPC driver
        passes (ilqr_core.py:457), is accepted as well.
        on_iteration(iteration, xs, us, J_opt, accepted, converged) is called after every
        iteration as in the reference; with a batch the arguments are arrays over the
        problems still running.  It makes the solve run one launch per iteration.
        Returns (xs [.., N+1, n], us [.., N, m]).
        """
        import torch
        if n_iterations is not None:
            n_interations = n_iterations
        n, m, N = self.dynamics.state_dimension, self.dynamics.control_dimension, self.N
        as_torch = _is_torch(x0) or _is_torch(u_seq_init)
        x0_nd = x0.dim() if _is_torch(x0) else np.ndim(x0)
        single = x0_nd == 1
        u_shape = tuple(u_seq_init.shape)
        assert u_shape[-2] == N, 'The length of control sequence should {}, yet is {} instead.'.format(N, u_shape[-2])
        x0_d = self._to_dev(x0, (-1, n))
        B = x0_d.shape[0]
        us_d = self._to_dev(u_seq_init, (B, N, m))
        delta, k, K = self._state_for_batch(B)
        M = self._model()
        kw = dict(tol=tolerance, line_search_steps=line_search_steps, scale=scale, mu_min=self.mu_min_,
                  mu_max=self.mu_max_ if self.mu_max_ else 0.0, delta0=self.delta_0_)

        if on_iteration is None:
            if isinstance(M, _hip.LargeModel):  # one problem per wavefront: a finished problem frees its wave
                res = self._solve(M, _hip.Opts(max_iter=n_interations, **kw), x0_d, us_d, delta=delta, k=k, K=K,
                                  workspace=getattr(self, "_large_workspace", None))
                self._large_workspace = res["workspace"]  # kept for the next call (re-solves of a closed loop)
            else:  # lane per problem: very large batches drop their finished problems between chunks of iterations
                from . import compaction
                res = compaction.solve_auto(lambda M_, O_, *a, **b: self._solve(M_, O_, *a, **b), M, kw, n_interations,
                                            x0_d, us_d, delta=delta, k=k, K=K)
            xs, us, J, mu = res["xs"], res["us"], res["J"], res["mu"]
            iters, status = res["iters"], res["status"]
        else:
            xs, us, J, mu, iters, status = self._solve_stepwise(M, kw, n_interations, x0_d, us_d, delta, k, K,
                                                                on_iteration, single, as_torch)
        # (no device synchronisation here: tensor callers get results ordered on the current stream,
        # numpy callers synchronise through the copies below)

        def out(t, shape=None):
            t = t[0] if single else t
            return t if as_torch else _to_numpy(t)
        # the gains stay on the device (they are the state of the next call); a numpy caller gets
        # them converted when k_ / K_ is first read
        self._gains = {"k": k[0] if single else k, "K": K[0] if single else K}
        self._gains_np = {}
        self._gains_as_numpy = not as_torch
        self.delta_ = float(delta[0]) if single else out(delta)
        self.mu_ = float(mu[0]) if single else out(mu)
        self.nominal_xs_, self.nominal_us_ = out(xs), out(us)
        self.J_ = float(J[0]) if single else out(J)
        self.iterations_ = int(iters[0]) if single else out(iters)
        self.status_ = int(status[0]) if single else out(status)
        return self.nominal_xs_, self.nominal_us_

    def _solve_stepwise(self, M, kw, max_iter, x0_d, us_d, delta, k, K, on_iteration, single, as_torch):
        """One launch per iteration so that the callback sees every iterate.  Each launch
        re-rolls the current controls (same inputs, same bits) and continues from the
        carried mu/delta/k/K, so the iterates equal those of the single-launch solve."""
        import torch
        B = x0_d.shape[0]
        dev = x0_d.device
        mu = torch.ones(B, dtype=torch.float64, device=dev)
        dJ = torch.zeros((2, B), dtype=torch.float64, device=dev)  # Delta_J1, Delta_J2: loop-carried (:278-279, :305)
        iters = torch.zeros(B, dtype=torch.int32, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        xs = torch.empty((B, self.N + 1, x0_d.shape[1]), dtype=torch.float64, device=dev)
        us = us_d.clone()
        J = torch.empty(B, dtype=torch.float64, device=dev)
        O = _hip.Opts(max_iter=1, mu_from_array=True, **kw)
        active_h = np.ones(B, dtype=bool)  # host copy of the running set: one device read-back per iteration
        ws = None

        def view(t):
            t = t[0] if single else t
            return t if as_torch else t.cpu().numpy()

        for it in range(max_iter):
            if not active_h.any():
                break
            everyone = bool(active_h.all())
            if everyone:
                # nothing has finished yet: the solver works on the full arrays in place
                res = self._solve(M, O, x0_d, us, delta=delta, k=k, K=K, mu=mu, trace=True, workspace=ws, dJ=dJ)
                ws = res["workspace"]
                xs, us, J = res["xs"], res["us"], res["J"]
                iters += 1
                status = res["status"]
                idx = None
            else:
                idx = torch.tensor(np.nonzero(active_h)[0], device=dev)
                d_i, k_i, K_i, mu_i = delta[idx].contiguous(), k[idx].contiguous(), K[idx].contiguous(), mu[idx].contiguous()
                dJ_i = dJ[:, idx].contiguous()
                M_i = M.subset(idx) if hasattr(M, "subset") else M  # (per-problem goals follow their problems)
                res = self._solve(M_i, O, x0_d[idx].contiguous(), us[idx].contiguous(), delta=d_i, k=k_i, K=K_i, mu=mu_i,
                                  trace=True, dJ=dJ_i)
                delta[idx], k[idx], K[idx], mu[idx], dJ[:, idx] = d_i, k_i, K_i, mu_i, dJ_i
                xs[idx], us[idx], J[idx] = res["xs"], res["us"], res["J"]
                iters[idx] += 1
                status[idx] = res["status"]
            flags = torch.stack([res["status"], res["trace"]["accepted"][0]]).cpu().numpy()
            conv_h = flags[0] == _hip.STATUS_CONVERGED
            stopped_h = flags[0] == _hip.STATUS_MU_MAX
            acc_h = flags[1] == 1
            # the reference leaves the loop on mu >= mu_max BEFORE the callback (ilqr_core.py:337-345)
            call_h = ~stopped_h
            if call_h.any():
                if call_h.all():
                    xs_c, us_c, J_c = res["xs"], res["us"], res["J"]
                else:
                    sel = torch.tensor(np.nonzero(call_h)[0], device=dev)
                    xs_c, us_c, J_c = res["xs"][sel], res["us"][sel], res["J"][sel]
                if single:
                    on_iteration(it, view(xs_c), view(us_c), float(J_c[0]), bool(acc_h[call_h][0]), bool(conv_h[call_h][0]))
                else:
                    acc_c, conv_c = acc_h[call_h], conv_h[call_h]
                    if as_torch:
                        acc_c, conv_c = torch.tensor(acc_c, device=dev), torch.tensor(conv_c, device=dev)
                    on_iteration(it, view(xs_c), view(us_c), view(J_c), acc_c, conv_c)
            active_h[np.nonzero(active_h)[0]] = ~(conv_h | stopped_h)
        return xs, us, J, mu, iters, status


class RecedingHorizonController(object):
    """Receding horizon controller for model predictive control (ilqr_core.py:400-466)."""

    def __init__(self, x0, u_seq_init, controller, env=None):
        self.xs_ = x0
        self.us_ = u_seq_init
        self.controller_ = controller
        self.env = env

    def set_state(self, x):
        """Set the current state [n] (or [B, n])."""
        self.xs_ = x

    def control(self, step_size=1, initial_n_iterations=200, subsequent_n_iterations=10, *args, **kwargs):
        """Generator of (xs[:step_size+1], us[:step_size]) per re-fit (ilqr_core.py:422-466).

        As in the reference the controller's state moves to where the model predicts the
        system will be after ``step_size`` steps; correct it with set_state() between
        iterations to close the loop on a real plant.  The reference as shipped raises
        TypeError here (it passes ``n_iterations=`` to a function whose parameter is spelt
        ``n_interations``); this is the behaviour that call intends.
        """
        n_iterations = initial_n_iterations
        while True:
            xs, us = self.controller_.update_u(self.xs_, self.us_, *args, n_iterations=n_iterations, **kwargs)
            batched = xs.ndim == 3
            if batched:
                self.xs_ = xs[:, step_size]
                yield xs[:, :step_size + 1], us[:, :step_size]
                us_start, us_end = us[:, step_size:], us[:, -step_size:]
                axis = 1
            else:
                self.xs_ = xs[step_size]
                yield xs[:step_size + 1], us[:step_size]
                us_start, us_end = us[step_size:], us[-step_size:]
                axis = 0
            # next seed: move along the current path and repeat its tail (ilqr_core.py:463-465)
            if _is_torch(us):
                import torch
                self.us_ = torch.cat([us_start, us_end], dim=axis)
            else:
                self.us_ = np.concatenate([us_start, us_end], axis=axis)
            n_iterations = subsequent_n_iterations
