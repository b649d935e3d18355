"""Dynamics interface of the solve path (mirror of ilqr/ilqr_dynamics.py:13-91).

The reference evaluates dynamics through Python callbacks (Theano thunks called
once per time step).  On the GPU the model is a device function selected by
``device_model``; a Dynamics subclass that does not name one cannot be solved by
``iLQR`` here (arbitrary Python callbacks cannot run inside a kernel).
"""
import abc


class Dynamics(abc.ABC):
    """Dynamics Model (ilqr/ilqr_dynamics.py:13-91)."""

    #: identifier of the device-side model (None: no kernel for this class)
    device_model = None

    @property
    @abc.abstractmethod
    def state_dimension(self):
        """State Dimension"""
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def control_dimension(self):
        """Control Dimension"""
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def has_hessians(self):
        """Whether the second order derivatives are available"""
        raise NotImplementedError

    @abc.abstractmethod
    def f(self, x, u, i):
        """Next state [state_dimension] from state x, control u at step i."""
        raise NotImplementedError

    @abc.abstractmethod
    def f_x(self, x, u, i):
        """df/dx [state_dimension, state_dimension]."""
        raise NotImplementedError

    @abc.abstractmethod
    def f_u(self, x, u, i):
        """df/du [state_dimension, control_dimension]."""
        raise NotImplementedError

    def f_xx(self, x, u, i):
        """d^2f/dx^2: second-order terms are not part of the iLQR path
        (dead code in the reference, ilqr_core.py:220-227)."""
        raise NotImplementedError

    def f_ux(self, x, u, i):
        raise NotImplementedError

    def f_uu(self, x, u, i):
        raise NotImplementedError


class FiniteDifferenceDynamics(Dynamics):
    """Finite-difference derivatives of a Python callable (ilqr/ilqr_dynamics.py:93-209), evaluated
    on the HOST with numpy: f_x[m][k] = (f(x + x_eps e_k, u)[m] - f(x, u)[m]) / x_eps (what
    ``scipy.optimize.approx_fprime`` computes), default increments sqrt(machine epsilon), and the
    reference's quirk that ``u_eps`` is only honoured when ``x_eps`` is given (:112-113).

    A Python callable cannot run inside a kernel, so ``iLQR`` does not accept this class; the same
    rule on the GPU is a model written as a C source with ``#define ILQR_USER_FD``
    (``ilqr_amd.user_model``).  The class is kept for inspecting models and checking Jacobians."""

    def __init__(self, f, state_dimension, control_dimension, x_eps=None, u_eps=None):
        import numpy as np
        self.f_ = f
        self.state_dimension_ = state_dimension
        self.control_dimension_ = control_dimension
        self.x_eps_ = x_eps if x_eps else np.sqrt(np.finfo(float).eps)
        self.u_eps_ = u_eps if x_eps else np.sqrt(np.finfo(float).eps)
        self.x_eps_hess_ = np.sqrt(self.x_eps_)
        self.u_eps_hess_ = np.sqrt(self.u_eps_)

    @property
    def state_dimension(self):
        return self.state_dimension_

    @property
    def control_dimension(self):
        return self.control_dimension_

    @property
    def has_hessians(self):
        return True

    def f(self, x, u, i):
        return self.f_(x, u, i)

    @staticmethod
    def _forward_differences(fun, z, eps):
        import numpy as np
        z = np.asarray(z, dtype=np.float64)
        f0 = np.asarray(fun(z), dtype=np.float64)
        cols = []
        for k in range(z.shape[0]):
            d = np.zeros_like(z)
            d[k] = eps
            cols.append((np.asarray(fun(z + d), dtype=np.float64) - f0) / eps)
        return np.stack(cols, axis=-1)

    def f_x(self, x, u, i):
        return self._forward_differences(lambda x_: self.f_(x_, u, i), x, self.x_eps_)

    def f_u(self, x, u, i):
        return self._forward_differences(lambda u_: self.f_(x, u_, i), u, self.u_eps_)

    def f_xx(self, x, u, i):
        # (:169-181: the reference differentiates f_x with u_eps_hess_, kept)
        return self._forward_differences(lambda x_: self.f_x(x_, u, i), x, self.u_eps_hess_)

    def f_ux(self, x, u, i):
        return self._forward_differences(lambda x_: self.f_u(x_, u, i), x, self.u_eps_hess_)

    def f_uu(self, x, u, i):
        return self._forward_differences(lambda u_: self.f_u(x, u_, i), u, self.u_eps_hess_)


class AutogradDynamics(Dynamics):
    """Placeholder for the reference's Theano-based class (ilqr/ilqr_dynamics.py:387-575): symbolic
    differentiation of a Python expression has no counterpart on the GPU path.  Write the model as a
    C source with its Jacobian (or ``#define ILQR_USER_FD``) and use ``ilqr_amd.user_model``."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "AutogradDynamics differentiates a Theano expression and evaluates it through Python callbacks; "
            "neither can run inside a kernel.  Use ilqr_amd.user_model (a C model source compiled with the solver).")

    state_dimension = control_dimension = has_hessians = None

    def f(self, x, u, i):
        raise NotImplementedError

    f_x = f_u = f
