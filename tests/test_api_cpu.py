"""CPU-side tests: host logic of the reference-API mirror, the C ABI surface, and the
rule that the product path has no CPU fallback."""
import ctypes
import os
import re

import numpy as np
import pytest

import ilqr_amd
from ilqr_amd import _hip
from ilqr_amd.cartpole import CartPoleDynamics
from ilqr_amd.ilqr_core import RecedingHorizonController, iLQR
from ilqr_amd.ilqr_cost import Cost, PathQRCost, QRCost
from ilqr_amd.ilqr_dynamics import Dynamics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ilqr_hip.h")).read()
    declared = set(re.findall(r"\b(ilqr_[a-z0-9_]+)\s*\(", header))
    declared -= {"ilqr_status"}
    assert len(declared) >= 14
    lib = ctypes.CDLL(_hip.LIB_PATH)  # loading needs no GPU
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(_hip.EXPORTS) == declared
    assert lib.ilqr_version() >= 100
    lib.ilqr_bm_ld.restype = ctypes.c_int64
    lib.ilqr_bm_ld.argtypes = [ctypes.c_int64]
    assert lib.ilqr_bm_ld(1) == 64 and lib.ilqr_bm_ld(64) == 64 and lib.ilqr_bm_ld(65) == 128
    lib.ilqr_status_string.restype = ctypes.c_char_p
    assert lib.ilqr_status_string(0) == b"ok" and b"argument" in lib.ilqr_status_string(-1)


def test_argument_checks_need_no_gpu():
    """Bad arguments are rejected before any device work."""
    lib = _hip.lib()
    M = _hip.Model(10, np.eye(5), np.eye(1), lower=[-1.0], upper=[1.0])
    assert lib.ilqr_solve_workspace_bytes(ctypes.byref(M.c), 100) == (2 * (11 * 5 + 10 + 10 + 50) * 128 * 8 + 2 * 128 * 4)
    assert lib.ilqr_rollout_cost_bm(ctypes.byref(M.c), ctypes.c_int64(4), ctypes.c_int64(64), None, None, None, None,
                                    None) == -1
    bad = _hip.Model(10, np.eye(5), np.eye(1), lower=[-1.0], upper=[1.0])
    bad.c.n = 4
    assert lib.ilqr_rollout_cost_bm(ctypes.byref(bad.c), ctypes.c_int64(4), ctypes.c_int64(64), None, None, None,
                                    None, None) == -2
    assert lib.ilqr_solve_workspace_bytes(ctypes.byref(bad.c), 100) == 0
    assert lib.ilqr_boxqp_batched(ctypes.c_int64(1), ctypes.c_int32(9), *([None] * 13)) == -1
    # large-state path: up to 32 states and 8 controls (padded layouts); missing arrays are argument errors
    L = _hip.LargeModelC()
    L.n, L.m, L.T = 32, 8, 20
    assert lib.ilqr_large_workspace_bytes(ctypes.byref(L), 10) == 0  # null matrices
    assert lib.ilqr_large_rollout_cost(ctypes.byref(L), ctypes.c_int64(4), None, None, None, None, None) == -1
    for name, kind in _hip.LargeModelC._fields_[6:]:
        if kind is ctypes.c_void_p:
            setattr(L, name, 4096)  # never dereferenced: the checks below fail first
    full = 10 * 8 * (21 * 32 + 20 * 8 + 20 * 8 + 20 * 8 * 32) + 256
    assert lib.ilqr_large_workspace_bytes(ctypes.byref(L), 10) == full
    assert lib.ilqr_large_rollout_cost(ctypes.byref(L), ctypes.c_int64(0), None, None, None, None, None) == -1
    L.n, L.m = 16, 4  # a smaller model keeps the full-size layouts
    assert lib.ilqr_large_workspace_bytes(ctypes.byref(L), 10) == full
    assert lib.ilqr_large_rollout_cost(ctypes.byref(L), ctypes.c_int64(4), None, None, None, None, None) == -1
    for n_bad, m_bad in ((33, 8), (32, 9), (0, 8), (32, 0)):
        L.n, L.m = n_bad, m_bad
        assert lib.ilqr_large_rollout_cost(ctypes.byref(L), ctypes.c_int64(4), None, None, None, None, None) == -2
    L.n, L.m = 32, 8
    L.goal_stride_x, L.goal_stride_u = 32, 0  # strides come in pairs
    assert lib.ilqr_large_rollout_cost(ctypes.byref(L), ctypes.c_int64(4), None, None, None, None, None) == -1
    L.goal_stride_x, L.goal_stride_u = 31, 8  # a goal does not fit its stride
    assert lib.ilqr_large_rollout_cost(ctypes.byref(L), ctypes.c_int64(4), None, None, None, None, None) == -1
    # cart-pole path: per-problem goals are for the solve entry points; the stage entry points refuse them
    pg = _hip.Model(10, np.eye(5), np.eye(1), lower=[-1.0], upper=[1.0])
    pg.c.goal_stride_x, pg.c.goal_stride_u = 5, 1
    assert lib.ilqr_rollout_cost_bm(ctypes.byref(pg.c), ctypes.c_int64(4), ctypes.c_int64(64), None, None, None,
                                    None, None) == -2
    assert lib.ilqr_solve_workspace_bytes(ctypes.byref(pg.c), 64) == \
        lib.ilqr_solve_workspace_bytes(ctypes.byref(_hip.Model(10, np.eye(5), np.eye(1), lower=[-1.0], upper=[1.0]).c), 64) \
        + 64 * 8 * 6
    pg.c.goal_stride_u = 0
    assert lib.ilqr_solve_workspace_bytes(ctypes.byref(pg.c), 64) == 0
    assert lib.ilqr_version() >= 102


def test_reference_import_names():
    import ilqr
    from ilqr import iLQR as A
    from ilqr.cartpole import CartPoleDynamics as B
    from ilqr.ilqr_cost import QRCost as C_
    from ilqr.ilqr_boxQP import boxQP  # noqa: F401
    from ilqr.ilqr_core import RecedingHorizonController as D
    assert A is iLQR and B is CartPoleDynamics and C_ is QRCost and D is RecedingHorizonController
    assert ilqr.iLQR is ilqr_amd.iLQR


def test_solver_construction_and_attributes():
    dyn = CartPoleDynamics(0.05)
    cost = QRCost(100 * np.eye(5), np.array([[5.0]]), Q_terminal=100 * np.eye(5),
                  x_goal=dyn.augment_state(np.zeros(4)))
    s = iLQR(dyn, cost, 100)
    assert (s.N, s.control_limit, s.mu_min_, s.mu_max_, s.mu_, s.delta_0_, s.delta_) == (100, True, 1e-6, 1e10, 1.0,
                                                                                          2.0, 2.0)
    assert s.k_.shape == (100, 1) and s.K_.shape == (100, 1, 5) and not s.k_.any() and not s.K_.any()
    assert np.array_equal(s.lower_bound, [-5.0]) and np.array_equal(s.upper_bound, [5.0])
    assert iLQR(dyn, cost).N == 15
    assert isinstance(dyn, Dynamics) and isinstance(cost, Cost)
    assert (dyn.state_dimension, dyn.control_dimension, dyn.has_hessians) == (5, 1, False)
    with pytest.raises(NotImplementedError):
        iLQR(dyn, cost, 10, control_limit=False)
    with pytest.raises(NotImplementedError):
        iLQR(dyn, cost, 10, use_second_order=True)

    class Py(Dynamics):
        state_dimension = control_dimension = 1
        has_hessians = False

        def f(self, x, u, i):
            return x

        f_x = f_u = f
    with pytest.raises(NotImplementedError):
        iLQR(Py(), cost, 10)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    dyn = CartPoleDynamics(0.05)
    s = iLQR(dyn, QRCost(np.eye(5), np.eye(1)), 10)
    with pytest.raises(_hip.IlqrHipError):
        s.update_u(np.zeros(5), np.zeros((10, 1)))
    from ilqr_amd.ilqr_boxQP import boxQP
    with pytest.raises(_hip.IlqrHipError):
        boxQP(np.eye(2), np.ones(2), -np.ones(2), np.ones(2), np.zeros(2))


def test_state_augmentation_round_trip():
    r = np.random.RandomState(0)
    st = r.uniform(-3, 3, (20, 4))
    aug = CartPoleDynamics.augment_state(st)
    assert aug.shape == (20, 5)
    np.testing.assert_allclose(aug[:, 2] ** 2 + aug[:, 3] ** 2, 1.0, rtol=1e-15)
    np.testing.assert_allclose(CartPoleDynamics.reduce_state(aug), st, rtol=1e-14, atol=1e-14)
    one = CartPoleDynamics.augment_state(np.array([0.0, 0.0, np.pi, 0.0]))
    assert one.shape == (5,) and one[3] == -1.0
    assert CartPoleDynamics.reduce_state(one).shape == (4,)


def test_qrcost_host_formulas(golden_dir):
    """Value with the 1/2, gradients without (ilqr_cost.py:547 vs :569)."""
    r = np.random.RandomState(1)
    Q, R = r.randn(5, 5), r.randn(1, 1)
    c = QRCost(Q, R, Q_terminal=2 * Q, x_goal=r.randn(5), u_goal=r.randn(1))
    x, u = r.randn(5), r.randn(1)
    xd, ud = x - c.x_goal, u - c.u_goal
    assert np.isclose(c.l(x, u, 0), 0.5 * xd @ Q @ xd + 0.5 * ud @ R @ ud)
    assert np.isclose(c.l(x, None, 3, terminal=True), xd @ Q @ xd)
    np.testing.assert_allclose(c.l_x(x, u, 0), xd @ (Q + Q.T))
    np.testing.assert_allclose(c.l_u(x, u, 0), ud @ (R + R.T))
    assert np.array_equal(c.l_xx(x, u, 0), Q + Q.T) and np.array_equal(c.l_xx(x, u, 0, True), 2 * (Q + Q.T))
    assert c.l_ux(x, u, 0).shape == (1, 5) and not c.l_ux(x, u, 0).any()
    assert not c.l_uu(x, u, 0, terminal=True).any() and not c.l_u(x, u, 0, terminal=True).any()
    with pytest.raises(AssertionError):
        QRCost(np.eye(5), np.eye(1), Q_terminal=np.eye(4))


def test_pathqrcost_host_formulas_against_reference_values(golden_dir):
    """PathQRCost: time-indexed goals, value WITHOUT the 1/2 (ilqr_cost.py:695-701); numbers
    returned by the reference's own class are in the fixture."""
    g = np.load(golden_dir + "/ref_pathqr_solve.npz")
    N = int(g["N"])
    c = PathQRCost(g["Q"], g["R"], g["x_path"], g["u_path"], Q_terminal=g["Qt"])
    for i in range(int(g["n_problems"])):
        p = "p%d_" % i
        xs, us = g[p + "xs"], g[p + "us"]
        np.testing.assert_allclose(c.l(xs[3], us[3], 3), g[p + "l0"], rtol=1e-13)
        np.testing.assert_allclose(c.l(xs[N], None, N, terminal=True), g[p + "lT"], rtol=1e-13)
        np.testing.assert_allclose(c.l_x(xs[3], us[3], 3), g[p + "lx0"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(c.l_u(xs[3], us[3], 3), g[p + "lu0"], rtol=1e-13, atol=1e-13)
    xd = xs[3] - g["x_path"][3]
    assert np.isclose(c.l(xs[3], us[3], 3), xd @ g["Q"] @ xd + (us[3] - g["u_path"][3]) @ g["R"] @ (us[3] - g["u_path"][3]))
    assert np.array_equal(c.l_xx(xs[0], us[0], 0), g["Q"] + g["Q"].T) and not c.l_ux(xs[0], us[0], 0).any()
    # u_path=None: zeros (the reference raises TypeError at ilqr_cost.py:663)
    z = PathQRCost(g["Q"], g["R"], g["x_path"])
    assert z.u_path.shape == (N, 1) and not z.u_path.any()
    with pytest.raises(AssertionError):
        PathQRCost(g["Q"], g["R"], g["x_path"], g["u_path"][:-1])


def test_receding_horizon_shift_logic():
    """The generator's bookkeeping (ilqr_core.py:456-466) with a stub controller."""
    class Stub:
        def __init__(self):
            self.calls = []

        def update_u(self, x, us, n_iterations=None):
            self.calls.append((np.array(x), np.array(us), n_iterations))
            T = us.shape[-2]
            if us.ndim == 3:
                xs = np.cumsum(np.concatenate([x[:, None, :], np.repeat(us, x.shape[-1], axis=-1)], 1), 1)
            else:
                xs = np.cumsum(np.vstack([x[None], np.repeat(us, x.shape[-1], axis=-1)]), 0)
            assert xs.shape[-2] == T + 1
            return xs, us + 1.0
    for batched in (False, True):
        st = Stub()
        x0 = np.zeros((3, 2)) if batched else np.zeros(2)
        us0 = np.arange(5.0).reshape(5, 1) * (np.ones((3, 1, 1)) if batched else 1.0)
        gen = RecedingHorizonController(x0, us0, st, None).control(step_size=2, initial_n_iterations=7,
                                                                   subsequent_n_iterations=3)
        xs, us = next(gen)
        assert xs.shape[-2] == 3 and us.shape[-2] == 2
        xs, us = next(gen)
        assert [c[2] for c in st.calls] == [7, 3]
        first_us = st.calls[0][1] + 1.0
        expect = np.concatenate([first_us[..., 2:, :], first_us[..., -2:, :]], axis=-2)
        assert np.array_equal(st.calls[1][1], expect)


def test_workload_definitions_agree_with_the_checkers():
    """The benchmark workloads are defined in the package (and bench.py); the CPU checker keeps
    its own copy of the same definitions: both must produce identical arrays."""
    from oracle import c_oracle as co
    from ilqr_amd.synthetic import SyntheticWorkload
    import bench
    wl, Mo = SyntheticWorkload(T=30), co.synthetic_model(T=30)
    for a, b in ((wl.A, Mo.A), (wl.B, Mo.B), (wl.W, Mo.W), (wl.Q, Mo.Q), (wl.R, Mo.R), (wl.Qt, Mo.Qt),
                 (wl.lower, Mo.lower), (wl.upper, Mo.upper)):
        assert np.array_equal(a, b)
    assert (wl.dt, wl.sig_c) == (Mo.dt, Mo.sig_c)
    for a, b in zip(wl.batch(7), co.synthetic_batch(7, T=30)):
        assert np.array_equal(a, b)
    for a, b in zip(bench.make_batch(9, 0), co.cartpole_batch(9)):
        assert np.array_equal(a, b)


def test_finite_difference_dynamics_mirror():
    """ilqr/ilqr_dynamics.py:93-167 on the host: forward differences with sqrt(eps) increments, the
    u_eps quirk (:113), and the solver's refusal of Python-callback models."""
    from ilqr_amd.ilqr_dynamics import AutogradDynamics, FiniteDifferenceDynamics
    A = np.array([[1.0, 0.1], [-0.2, 0.9]])
    Bm = np.array([[0.0], [0.5]])
    f = lambda x, u, i: A.dot(x) + Bm.dot(u) + 0.01 * np.array([np.sin(x[0]), x[1] * u[0]])  # noqa: E731
    d = FiniteDifferenceDynamics(f, 2, 1)
    assert d.x_eps_ == np.sqrt(np.finfo(float).eps) == 2.0 ** -26
    assert FiniteDifferenceDynamics(f, 2, 1, u_eps=1e-3).u_eps_ == 2.0 ** -26        # ignored without x_eps
    assert FiniteDifferenceDynamics(f, 2, 1, x_eps=1e-4, u_eps=1e-3).u_eps_ == 1e-3
    x, u = np.array([0.3, -0.7]), np.array([0.4])
    fx, fu = d.f_x(x, u, 0), d.f_u(x, u, 0)
    assert fx.shape == (2, 2) and fu.shape == (2, 1)
    col0 = (f(x + np.array([d.x_eps_, 0.0]), u, 0) - f(x, u, 0)) / d.x_eps_
    assert np.array_equal(fx[:, 0], col0)
    np.testing.assert_allclose(fx, A + 0.01 * np.array([[np.cos(0.3), 0.0], [0.0, 0.4]]), atol=1e-6)
    np.testing.assert_allclose(fu, Bm + 0.01 * np.array([[0.0], [-0.7]]), atol=1e-6)
    assert d.f_xx(x, u, 0).shape == (2, 2, 2) and d.has_hessians
    with pytest.raises(NotImplementedError):
        iLQR(d, QRCost(np.eye(2), np.eye(1)), 5)
    with pytest.raises(NotImplementedError):
        AutogradDynamics(None, None, None)
    from ilqr_amd.ilqr_cost import AutogradCost, FiniteDifferenceCost
    for callback_cost in (AutogradCost, FiniteDifferenceCost):  # importable, refused with an explanation
        with pytest.raises(NotImplementedError, match="QRCost / PathQRCost"):
            callback_cost(None, None, 2, 1)
