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
    (l = x'Qx/2 + u'Ru/2 but l_x = x'(Q+Q'), l_xx = Q+Q', l_uu = R+R')."""

    def __init__(self, Q, R, Q_terminal=None, x_goal=None, u_goal=None):
        self.Q = np.array(Q, dtype=np.float64)
        self.R = np.array(R, dtype=np.float64)
        self.Q_terminal = self.Q if Q_terminal is None else np.array(Q_terminal, dtype=np.float64)
        self.x_goal = np.zeros(self.Q.shape[0]) if x_goal is None else np.array(x_goal, dtype=np.float64)
        self.u_goal = np.zeros(self.R.shape[0]) if u_goal is None else np.array(u_goal, dtype=np.float64)

        assert self.Q.shape == self.Q_terminal.shape, "Q & Q_terminal mismatch"
        assert self.Q.shape[0] == self.Q.shape[1], "Q must be square"
        assert self.R.shape[0] == self.R.shape[1], "R must be square"
        assert self.Q.shape[0] == self.x_goal.shape[0], "Q & x_goal mismatch"
        assert self.R.shape[0] == self.u_goal.shape[0], "R & u_goal mismatch"

        self._Q_plus_Q_T = self.Q + self.Q.T
        self._R_plus_R_T = self.R + self.R.T
        self._Q_plus_Q_T_terminal = self.Q_terminal + self.Q_terminal.T
        super().__init__()

    def l(self, x, u, i, terminal=False):
        Q = self.Q_terminal if terminal else self.Q
        x_diff = x - self.x_goal
        squared_x_cost = x_diff.T.dot(Q).dot(x_diff) * 0.5
        if terminal:
            return squared_x_cost
        u_diff = u - self.u_goal
        return squared_x_cost + u_diff.T.dot(self.R).dot(u_diff) * 0.5

    def l_x(self, x, u, i, terminal=False):
        Q_plus_Q_T = self._Q_plus_Q_T_terminal if terminal else self._Q_plus_Q_T
        return (x - self.x_goal).T.dot(Q_plus_Q_T)

    def l_u(self, x, u, i, terminal=False):
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
