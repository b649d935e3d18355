"""Cost interface and the quadratic costs (mirror of ilqr/ilqr_cost.py).

The methods below are host-side numpy evaluations with the reference's exact
formulas (they are what a user calls to inspect a cost); the solver evaluates the
same expressions inside its kernels from the matrices stored here.
"""
import abc

import numpy as np


class Cost(abc.ABC):
    """Instantaneous Cost Function (ilqr/ilqr_cost.py:12-91).

    The terminal cost is a function of x and i only; the non-terminal cost of x, u and i.
    """

    @abc.abstractmethod
    def l(self, x, u, i, terminal=False):
        raise NotImplementedError

    @abc.abstractmethod
    def l_x(self, x, u, i, terminal=False):
        raise NotImplementedError

    @abc.abstractmethod
    def l_u(self, x, u, i, terminal=False):
        raise NotImplementedError

    @abc.abstractmethod
    def l_xx(self, x, u, i, terminal=False):
        raise NotImplementedError

    @abc.abstractmethod
    def l_ux(self, x, u, i, terminal=False):
        raise NotImplementedError

    @abc.abstractmethod
    def l_uu(self, x, u, i, terminal=False):
        raise NotImplementedError


class QRCost(Cost):
    """Quadratic Regulator Instantaneous Cost (ilqr/ilqr_cost.py:486-632).

    As in the reference the VALUE carries a factor 1/2 that the derivatives do not
    (l = x'Qx/2 + u'Ru/2 but l_x = x'(Q+Q'), l_xx = Q+Q', l_uu = R+R').

    Batched extension: ``x_goal`` [B, n] (and ``u_goal`` [B, m]) gives every problem of a batched solve its own
    goal, as B solver objects of the reference, each with its own QRCost, would have; the host-side methods
    below then evaluate problem 0's goal unless given the goal row explicitly through ``for_problem(b)``."""

    def __init__(self, Q, R, Q_terminal=None, x_goal=None, u_goal=None):
        self.Q = np.array(Q, dtype=np.float64)
        self.R = np.array(R, dtype=np.float64)
        self.Q_terminal = self.Q if Q_terminal is None else np.array(Q_terminal, dtype=np.float64)
        self.x_goal = np.zeros(self.Q.shape[0]) if x_goal is None else np.array(x_goal, dtype=np.float64)
        self.u_goal = np.zeros(self.R.shape[0]) if u_goal is None else np.array(u_goal, dtype=np.float64)
        if self.x_goal.ndim == 2 and self.u_goal.ndim == 1:  # one state goal per problem, one control goal for all
            self.u_goal = np.broadcast_to(self.u_goal, (self.x_goal.shape[0], self.u_goal.shape[0])).copy()
        elif self.u_goal.ndim == 2 and self.x_goal.ndim == 1:
            self.x_goal = np.broadcast_to(self.x_goal, (self.u_goal.shape[0], self.x_goal.shape[0])).copy()

        assert self.Q.shape == self.Q_terminal.shape, "Q & Q_terminal mismatch"
        assert self.Q.shape[0] == self.Q.shape[1], "Q must be square"
        assert self.R.shape[0] == self.R.shape[1], "R must be square"
        assert self.Q.shape[0] == self.x_goal.shape[-1], "Q & x_goal mismatch"
        assert self.R.shape[0] == self.u_goal.shape[-1], "R & u_goal mismatch"
        assert self.x_goal.ndim == self.u_goal.ndim and self.x_goal.shape[:-1] == self.u_goal.shape[:-1], \
            "x_goal & u_goal batch mismatch"

        self._Q_plus_Q_T = self.Q + self.Q.T
        self._R_plus_R_T = self.R + self.R.T
        self._Q_plus_Q_T_terminal = self.Q_terminal + self.Q_terminal.T
        super().__init__()

    def for_problem(self, b):
        """The QRCost of problem ``b`` of a batch with per-problem goals (a plain single-goal cost)."""
        if self.x_goal.ndim == 1:
            return self
        return QRCost(self.Q, self.R, self.Q_terminal, self.x_goal[b], self.u_goal[b])

    def l(self, x, u, i, terminal=False):
        if self.x_goal.ndim == 2:
            return self.for_problem(0).l(x, u, i, terminal)
        Q = self.Q_terminal if terminal else self.Q
        x_diff = x - self.x_goal
        squared_x_cost = x_diff.T.dot(Q).dot(x_diff) * 0.5
        if terminal:
            return squared_x_cost
        u_diff = u - self.u_goal
        return squared_x_cost + u_diff.T.dot(self.R).dot(u_diff) * 0.5

    def l_x(self, x, u, i, terminal=False):
        if self.x_goal.ndim == 2:
            return self.for_problem(0).l_x(x, u, i, terminal)
        Q_plus_Q_T = self._Q_plus_Q_T_terminal if terminal else self._Q_plus_Q_T
        return (x - self.x_goal).T.dot(Q_plus_Q_T)

    def l_u(self, x, u, i, terminal=False):
        if self.x_goal.ndim == 2:
            return self.for_problem(0).l_u(x, u, i, terminal)
        if terminal:
            return np.zeros_like(self.u_goal)
        return (u - self.u_goal).T.dot(self._R_plus_R_T)

    def l_xx(self, x, u, i, terminal=False):
        return self._Q_plus_Q_T_terminal if terminal else self._Q_plus_Q_T

    def l_ux(self, x, u, i, terminal=False):
        return np.zeros((self.R.shape[0], self.Q.shape[0]))

    def l_uu(self, x, u, i, terminal=False):
        if terminal:
            return np.zeros_like(self.R)
        return self._R_plus_R_T


class PathQRCost(Cost):
    """Quadratic Regulator Instantaneous Cost for trajectory following
    (ilqr/ilqr_cost.py:635-770): goals x_path[i], u_path[i] indexed by the time step.

    As in the reference the VALUE has no factor 1/2 here (l = x'Qx + u'Ru, :695-701)
    while the derivatives are those of QRCost (l_x = x'(Q+Q'), ...).  ``x_path`` [N+1, n] is one
    path shared by every problem of a batch; ``x_path`` [B, N+1, n] (with ``u_path`` [B, N, m]) gives every
    problem of a batched solve its own path, as B solver objects of the reference would have
    (``for_problem(b)`` is problem b's plain cost for host-side evaluation).

    ``u_path=None``: the reference crashes there (``np.zeros(path_length - 1,
    action_size)``, :663, is a TypeError); the zeros it meant to build are used."""

    def __init__(self, Q, R, x_path, u_path=None, Q_terminal=None):
        self.Q = np.array(Q, dtype=np.float64)
        self.R = np.array(R, dtype=np.float64)
        self.x_path = np.array(x_path, dtype=np.float64)

        state_size = self.Q.shape[0]
        action_size = self.R.shape[0]
        path_length = self.x_path.shape[-2]

        self.Q_terminal = self.Q if Q_terminal is None else np.array(Q_terminal, dtype=np.float64)
        if u_path is None:
            self.u_path = np.zeros(self.x_path.shape[:-2] + (path_length - 1, action_size))
        else:
            self.u_path = np.array(u_path, dtype=np.float64)

        assert self.Q.shape == self.Q_terminal.shape, "Q & Q_terminal mismatch"
        assert self.Q.shape[0] == self.Q.shape[1], "Q must be square"
        assert self.R.shape[0] == self.R.shape[1], "R must be square"
        assert state_size == self.x_path.shape[-1], "Q & x_path mismatch"
        assert action_size == self.u_path.shape[-1], "R & u_path mismatch"
        assert path_length == self.u_path.shape[-2] + 1, "x_path must be 1 longer than u_path"
        assert self.x_path.ndim == self.u_path.ndim and self.x_path.shape[:-2] == self.u_path.shape[:-2], \
            "x_path & u_path batch mismatch"

        self._Q_plus_Q_T = self.Q + self.Q.T
        self._R_plus_R_T = self.R + self.R.T
        self._Q_plus_Q_T_terminal = self.Q_terminal + self.Q_terminal.T
        self._device_paths = {}
        super().__init__()

    def device_paths(self, device):
        """x_path, u_path as fp64 tensors on ``device``.  The tensors are allocated once per device and keep
        their addresses; their CONTENTS follow the host arrays: a caller that shifts ``x_path`` / ``u_path``
        between two ``update_u`` calls (receding-horizon tracking, ilqr_core.py:437-438) gets the new path on
        the device at the next call."""
        import torch
        key = str(device)
        x_h = np.ascontiguousarray(self.x_path, dtype=np.float64)
        u_h = np.ascontiguousarray(self.u_path, dtype=np.float64)
        entry = self._device_paths.get(key)
        if entry is None or entry[2].shape != x_h.shape or entry[3].shape != u_h.shape:
            entry = [torch.tensor(x_h, device=device).contiguous(), torch.tensor(u_h, device=device).contiguous(),
                     x_h.copy(), u_h.copy()]
            self._device_paths[key] = entry
        else:
            if not np.array_equal(entry[2], x_h):
                entry[0].copy_(torch.from_numpy(x_h))
                entry[2] = x_h.copy()
            if not np.array_equal(entry[3], u_h):
                entry[1].copy_(torch.from_numpy(u_h))
                entry[3] = u_h.copy()
        return entry[0], entry[1]

    def for_problem(self, b):
        """The PathQRCost of problem ``b`` of a batch with per-problem paths."""
        if self.x_path.ndim == 2:
            return self
        return PathQRCost(self.Q, self.R, self.x_path[b], self.u_path[b], self.Q_terminal)

    def l(self, x, u, i, terminal=False):
        if self.x_path.ndim == 3:
            return self.for_problem(0).l(x, u, i, terminal)
        Q = self.Q_terminal if terminal else self.Q
        x_diff = x - self.x_path[i]
        squared_x_cost = x_diff.T.dot(Q).dot(x_diff)
        if terminal:
            return squared_x_cost
        u_diff = u - self.u_path[i]
        return squared_x_cost + u_diff.T.dot(self.R).dot(u_diff)

    def l_x(self, x, u, i, terminal=False):
        Q_plus_Q_T = self._Q_plus_Q_T_terminal if terminal else self._Q_plus_Q_T
        return (x - self.x_path[i]).T.dot(Q_plus_Q_T)

    def l_u(self, x, u, i, terminal=False):
        if terminal:
            return np.zeros_like(self.u_path)
        return (u - self.u_path[i]).T.dot(self._R_plus_R_T)

    def l_xx(self, x, u, i, terminal=False):
        return self._Q_plus_Q_T_terminal if terminal else self._Q_plus_Q_T

    def l_ux(self, x, u, i, terminal=False):
        return np.zeros((self.R.shape[0], self.Q.shape[0]))

    def l_uu(self, x, u, i, terminal=False):
        if terminal:
            return np.zeros_like(self.R)
        return self._R_plus_R_T


class _CallbackCost(Cost):
    """Costs defined by Python callables (ilqr/ilqr_cost.py:93-483) are evaluated once per time step
    through the interpreter, which cannot happen inside a kernel; the device solver evaluates the
    quadratic families above (``QRCost``, ``PathQRCost``) in registers.  The names are kept so that
    code importing them fails with an explanation instead of an ImportError."""

    _what = "a cost given as Python callables"

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "%s is not available on the GPU path: the solve kernels evaluate QRCost / PathQRCost only "
            "(a tracking problem with time-varying goals is a PathQRCost)." % self._what)

    def l(self, x, u, i, terminal=False):
        raise NotImplementedError

    l_x = l_u = l_xx = l_ux = l_uu = l


class FiniteDifferenceCost(_CallbackCost):
    """ilqr/ilqr_cost.py:93-205."""
    _what = "FiniteDifferenceCost (numerical derivatives of a Python cost function)"


class AutogradCost(_CallbackCost):
    """ilqr/ilqr_cost.py:207-483."""
    _what = "AutogradCost (symbolic derivatives of a Theano cost expression)"
