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
