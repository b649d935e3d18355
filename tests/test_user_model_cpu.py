"""User-defined dynamics, CPU side: the model headers under tests/models compile for the host
(oracle/user_model_shim.c), and the oracle run on the cart-pole written as a user model is the
oracle run on its built-in cart-pole, bit for bit (same expressions, same solver code)."""
import ctypes as C
import os

import numpy as np

from oracle import c_oracle as co

MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")


def cartpole_as_user(T):
    lib = co.build_user_model(os.path.join(MODELS, "cartpole_user.h"), "cartpole_user")
    Mb = co.cartpole_model(T=T)
    Mu = co.user_model(lib, [0.05, 1.0, 0.1, 1.0, 9.8], T, Mb.Q, Mb.R, Mb.Qt, Mb.x_goal, None, Mb.lower, Mb.upper)
    return Mb, Mu


def pendulum(T):
    lib = co.build_user_model(os.path.join(MODELS, "pendulum_user.h"), "pendulum_user")
    Q = np.diag([20.0, 20.0, 1.0])
    return co.user_model(lib, [0.05, 9.81, 0.2, 1.5], T, Q, [[0.5]], 100.0 * np.eye(3), [0.0, 1.0, 0.0], None,
                         [-2.0], [2.0])


def pendulum_fd(T):
    """The same pendulum with forward-difference derivatives (FiniteDifferenceDynamics' rule)."""
    lib = co.build_user_model(os.path.join(MODELS, "pendulum_fd_user.h"), "pendulum_fd_user")
    Ma = pendulum(T)
    return co.user_model(lib, Ma.user_params, T, Ma.Q, Ma.R, Ma.Qt, Ma.x_goal, None, Ma.lower, Ma.upper)


ROTOR_PARAMS = [0.05, 9.81, 1.0, 4.0, 0.1]


def rotor(T):
    """The two-control model (tests/models/planar_rotor_user.h): fly to the origin, level, at rest."""
    lib = co.build_user_model(os.path.join(MODELS, "planar_rotor_user.h"), "planar_rotor_user")
    Q = np.diag([10.0, 10.0, 5.0, 5.0, 1.0, 1.0, 1.0])
    R = np.array([[0.2, 0.05], [0.05, 0.2]])
    xg = [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
    return co.user_model(lib, ROTOR_PARAMS, T, Q, R, 50.0 * np.eye(7), xg, None, [-3.0, -3.0], [3.0, 3.0])


def rotor_fd(T):
    """The same vehicle with forward-difference derivatives (two controls)."""
    lib = co.build_user_model(os.path.join(MODELS, "planar_rotor_fd_user.h"), "planar_rotor_fd_user")
    Ma = rotor(T)
    return co.user_model(lib, Ma.user_params, T, Ma.Q, Ma.R, Ma.Qt, Ma.x_goal, None, Ma.lower, Ma.upper)


def rotor_batch(B, T, seed=7):
    r = np.random.RandomState(seed)
    phi = r.uniform(-1.0, 1.0, B)
    x0 = np.stack([r.uniform(-2, 2, B), r.uniform(-2, 2, B), np.sin(phi), np.cos(phi), r.uniform(-1, 1, B),
                   r.uniform(-1, 1, B), r.uniform(-1, 1, B)], axis=1)
    return x0, r.uniform(-1, 1, (B, T, 2))


CHAIN_PARAMS = [0.05, 4.0, 0.5, 0.5]


def chain(T):
    """The largest model the per-model solver takes (tests/models/mass_chain_user.h): 8 states, 4 controls;
    bring the four masses to rest at their unloaded positions, forces limited to +-0.6."""
    lib = co.build_user_model(os.path.join(MODELS, "mass_chain_user.h"), "mass_chain_user")
    Q = np.diag([8.0, 8.0, 8.0, 8.0, 1.0, 1.0, 1.0, 1.0])
    R = 0.1 * np.eye(4) + 0.02 * np.ones((4, 4))
    return co.user_model(lib, CHAIN_PARAMS, T, Q, R, 40.0 * np.eye(8), np.zeros(8), None, [-0.6] * 4, [0.6] * 4)


def chain_batch(B, T, seed=5):
    r = np.random.RandomState(seed)
    return np.concatenate([r.uniform(-0.8, 0.8, (B, 4)), r.uniform(-0.5, 0.5, (B, 4))], axis=1), r.uniform(-1, 1, (B, T, 4))


def pendulum_batch(B, T, seed=3):
    r = np.random.RandomState(seed)
    th, w = r.uniform(-np.pi, np.pi, B), r.uniform(-1, 1, B)
    return np.stack([np.sin(th), np.cos(th), w], axis=1), r.uniform(-1, 1, (B, T, 1))


def test_user_cartpole_is_the_builtin_cartpole():
    T, B = 40, 12
    Mb, Mu = cartpole_as_user(T)
    x0, us0 = co.cartpole_batch(B, T=T)
    fb, fub = co.dynamics(Mb, x0[0], us0[0, 0]), co.dynamics(Mu, x0[0], us0[0, 0])
    for a, b in zip(fb, fub):
        assert np.array_equal(a, b)
    a = co.solve(Mb, co.Opts(max_iter=30), x0, us0)
    b = co.solve(Mu, co.Opts(max_iter=30), x0, us0)
    for key in ("xs", "us", "J", "mu", "delta", "k", "K", "iters", "status"):
        assert np.array_equal(a[key], b[key]), key


def test_pendulum_jacobian_matches_finite_differences_and_solve_descends():
    T, B = 60, 6
    M = pendulum(T)
    x0, us0 = pendulum_batch(B, T)
    f, fx, fu = co.dynamics(M, x0[0], us0[0, 0])
    eps = 1e-6
    for j in range(3):
        d = np.zeros(3)
        d[j] = eps
        fp, fm = co.dynamics(M, x0[0] + d, us0[0, 0])[0], co.dynamics(M, x0[0] - d, us0[0, 0])[0]
        np.testing.assert_allclose(fx[0][:, j], (fp[0] - fm[0]) / (2 * eps), atol=1e-7)
    fp, fm = co.dynamics(M, x0[0], us0[0, 0] + eps)[0], co.dynamics(M, x0[0], us0[0, 0] - eps)[0]
    np.testing.assert_allclose(fu[0][:, 0], (fp[0] - fm[0]) / (2 * eps), atol=1e-7)
    J0 = co.rollout_cost(M, x0, us0)[1]
    out = co.solve(M, co.Opts(max_iter=40), x0, us0)
    # (the reference's forward pass does not clip u + alpha k + K dx, so |u| may exceed the bound)
    assert (out["J"] < J0).all() and (out["trace"]["n_clamped"] > 0).any()


def test_two_control_model_jacobian_and_solve():
    T, B = 40, 6
    M = rotor(T)
    assert (M.n, M.m) == (7, 2)
    x0, us0 = rotor_batch(B, T)
    f, fx, fu = co.dynamics(M, x0[0], us0[0, 0])
    eps = 1e-6
    for j in range(7):
        d = np.zeros(7)
        d[j] = eps
        col = (co.dynamics(M, x0[0] + d, us0[0, 0])[0][0] - co.dynamics(M, x0[0] - d, us0[0, 0])[0][0]) / (2 * eps)
        np.testing.assert_allclose(fx[0][:, j], col, atol=1e-7)
    for j in range(2):
        d = np.zeros(2)
        d[j] = eps
        col = (co.dynamics(M, x0[0], us0[0, 0] + d)[0][0] - co.dynamics(M, x0[0], us0[0, 0] - d)[0][0]) / (2 * eps)
        np.testing.assert_allclose(fu[0][:, j], col, atol=1e-7)
    J0 = co.rollout_cost(M, x0, us0)[1]
    out = co.solve(M, co.Opts(max_iter=40), x0, us0)
    assert (out["J"] < 0.5 * J0).all() and (out["trace"]["n_clamped"] > 0).any()


def test_largest_model_jacobian_and_solve():
    """8 states, 4 controls: the analytic Jacobian against central differences, and the oracle's solve
    descends with limits active."""
    T, B = 30, 5
    M = chain(T)
    assert (M.n, M.m) == (8, 4)
    x0, us0 = chain_batch(B, T)
    f, fx, fu = co.dynamics(M, x0[0], us0[0, 0])
    eps = 1e-6
    for j in range(8):
        d = np.zeros(8)
        d[j] = eps
        col = (co.dynamics(M, x0[0] + d, us0[0, 0])[0][0] - co.dynamics(M, x0[0] - d, us0[0, 0])[0][0]) / (2 * eps)
        np.testing.assert_allclose(fx[0][:, j], col, atol=1e-7)
    for j in range(4):
        d = np.zeros(4)
        d[j] = eps
        col = (co.dynamics(M, x0[0], us0[0, 0] + d)[0][0] - co.dynamics(M, x0[0], us0[0, 0] - d)[0][0]) / (2 * eps)
        np.testing.assert_allclose(fu[0][:, j], col, atol=1e-7)
    J0 = co.rollout_cost(M, x0, us0)[1]
    out = co.solve(M, co.Opts(max_iter=40), x0, us0)
    assert (out["J"] < 0.9 * J0).all() and (out["trace"]["n_clamped"] > 0).any()


def test_forward_difference_model_follows_approx_fprime():
    """ilqr/ilqr_dynamics.py:145-167: row m of f_x is approx_fprime(x, f(.)[m], x_eps), i.e.
    (f(x + eps e_k) - f(x)) / eps with eps = sqrt(machine epsilon); checked against that formula
    evaluated with the oracle's own f, and against the analytic Jacobian."""
    T = 20
    Ma, Mf = pendulum(T), pendulum_fd(T)
    x0, us0 = pendulum_batch(5, T, seed=9)
    eps = np.sqrt(np.finfo(float).eps)
    for b in range(5):
        x, u = x0[b], us0[b, 0]
        f, fx, fu = co.dynamics(Mf, x, u)
        fa, fxa, fua = co.dynamics(Ma, x, u)
        assert np.array_equal(f, fa)
        for k in range(3):
            d = np.zeros(3)
            d[k] = eps
            col = (co.dynamics(Ma, x + d, u)[0][0] - fa[0]) / eps
            assert np.array_equal(fx[0][:, k], col)
        assert np.array_equal(fu[0][:, 0], (co.dynamics(Ma, x, u + eps)[0][0] - fa[0]) / eps)
        np.testing.assert_allclose(fx[0], fxa[0], atol=2e-7)
        np.testing.assert_allclose(fu[0], fua[0], atol=2e-7)
    a = co.solve(Ma, co.Opts(max_iter=30), x0, us0)
    b_ = co.solve(Mf, co.Opts(max_iter=30), x0, us0)
    np.testing.assert_allclose(b_["J"], a["J"], rtol=1e-3)


def test_forward_difference_model_with_two_controls():
    """ILQR_USER_FD on a model with a control vector: f_x, f_u within the truncation error of the analytic
    Jacobian of the same vehicle, and the solve descends."""
    T, B = 25, 4
    Ma, Mf = rotor(T), rotor_fd(T)
    assert (Mf.n, Mf.m) == (7, 2)
    x0, us0 = rotor_batch(B, T, seed=12)
    _, fx_a, fu_a = co.dynamics(Ma, x0[1], us0[1, 0])
    f, fx, fu = co.dynamics(Mf, x0[1], us0[1, 0])
    np.testing.assert_allclose(fx, fx_a, atol=2e-6)
    np.testing.assert_allclose(fu, fu_a, atol=2e-6)
    assert not np.array_equal(fx, fx_a)  # forward differences, not a copy of the analytic entries
    J0 = co.rollout_cost(Mf, x0, us0)[1]
    out = co.solve(Mf, co.Opts(max_iter=30), x0, us0)
    assert (out["J"] < 0.9 * J0).all()


def test_user_model_libraries_export_every_declared_symbol():
    import re
    from ilqr_amd import user_model as um
    root = os.path.dirname(MODELS.rstrip("/").rsplit("/", 1)[0])
    header = open(os.path.join(root, "include", "ilqr_user.h")).read()
    body = header[header.index("#ifndef ILQR_USER_H"):]
    declared = set(re.findall(r"\b(ilqr_user_[a-z0-9_]+)\s*\(", body))
    assert declared == {"ilqr_user_dims", "ilqr_user_controls", "ilqr_user_workspace_bytes", "ilqr_user_solve_batched"}
    for name, dims in (("cartpole_user", (5, 5)), ("pendulum_user", (3, 4)), ("pendulum_fd_user", (3, 4)),
                       ("planar_rotor_user", (7, 5)), ("mass_chain_user", (8, 4)),
                       ("planar_rotor_fd_user", (7, 5))):
        um.build_user_model(os.path.join(MODELS, name + ".h"), name)  # hipcc cross-compiles; no-op when up to date
        lib = C.CDLL(um.library_path(name))  # loading needs no GPU
        for sym in declared:
            assert hasattr(lib, sym), (name, sym)
        n, npar = C.c_int32(), C.c_int32()
        assert lib.ilqr_user_dims(C.byref(n), C.byref(npar)) == 0 and (n.value, npar.value) == dims
        assert lib.ilqr_user_controls() == {"planar_rotor_user": 2, "planar_rotor_fd_user": 2, "mass_chain_user": 4}.get(name, 1)


def test_large_user_model_library_and_its_jacobian():
    """A model with more than 8 states: the per-model build of the large-state path exports the ilqr_large_*
    entry points (loading needs no GPU) and refuses other dimensions; the model's Jacobian agrees with central
    differences of its f, and the oracle's solve on it descends."""
    import ctypes as C
    import re
    from ilqr_amd import _hip, user_model as um
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    um.build_large_user_model(os.path.join(MODELS, "long_chain_user.h"), "long_chain_user")
    lib = C.CDLL(um.large_library_path("long_chain_user"))
    header = open(os.path.join(root, "include", "ilqr_hip.h")).read()
    declared = set(re.findall(r"\b(ilqr_large_[a-z0-9_]+)\s*\(", header)) - {"ilqr_large_model"}
    assert {"ilqr_large_solve_batched", "ilqr_large_backward", "ilqr_large_forward_ls", "ilqr_large_rollout_cost",
            "ilqr_large_workspace_bytes"} <= declared
    for sym in declared | {"ilqr_large_user_dims"}:
        assert hasattr(lib, sym), sym
    assert um.large_user_dims(lib) == (20, 5, 4)
    L = _hip.LargeModelC()
    L.n, L.m, L.T = 32, 8, 10  # not this library's model
    assert lib.ilqr_large_rollout_cost(C.byref(L), C.c_int64(4), None, None, None, None, None) == -2
    L.n, L.m = 20, 5
    assert lib.ilqr_large_rollout_cost(C.byref(L), C.c_int64(4), None, None, None, None, None) == -1  # null arrays
    # the model source itself, on the CPU
    olib = co.build_user_model(os.path.join(MODELS, "long_chain_user.h"), "long_chain_user")
    p = np.array([0.05, 4.0, 6.0, 0.3])
    Mo = co.user_model(olib, p, 20, np.eye(20), 0.2 * np.eye(5), 10.0 * np.eye(20), None, None, [-1.5] * 5, [1.5] * 5)
    r = np.random.RandomState(3)
    x, u = r.uniform(-1, 1, 20), r.uniform(-1, 1, 5)
    f0, fx, fu = co.dynamics(Mo, x, u)
    h = 1e-6
    for j in range(20):
        d = np.zeros(20)
        d[j] = h
        num = (co.dynamics(Mo, x + d, u)[0][0] - co.dynamics(Mo, x - d, u)[0][0]) / (2 * h)
        assert np.allclose(fx[0][:, j], num, atol=1e-7)
    for j in range(5):
        d = np.zeros(5)
        d[j] = h
        num = (co.dynamics(Mo, x, u + d)[0][0] - co.dynamics(Mo, x, u - d)[0][0]) / (2 * h)
        assert np.allclose(fu[0][:, j], num, atol=1e-7)
    x0 = r.uniform(-0.5, 0.5, (3, 20))
    us0 = r.uniform(-0.5, 0.5, (3, 20, 5))
    out = co.solve(Mo, co.Opts(max_iter=10), x0, us0)
    J0 = co.rollout_cost(Mo, x0, us0)[1]
    assert (out["J"] < J0).all()
