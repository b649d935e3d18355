"""GPU parity of user-defined dynamics (per-model libraries built from tests/models/*.h by
__graft_entry__.build()).  The reference has no compiled-model mechanism: its counterpart is a
Python Dynamics object evaluated per step (ilqr/ilqr_dynamics.py:94-575).  Pinning: the cart-pole
written as a user model must reproduce the built-in cart-pole path, which the reference runs pin;
the pendulum is compared with the oracle's generic solver on the same model source.  Bar: ==."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import c_oracle as co  # noqa: E402
from test_user_model_cpu import (cartpole_as_user, chain, chain_batch, pendulum, pendulum_batch,  # noqa: E402
                                 pendulum_fd, rotor, rotor_batch, rotor_fd)


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ilqr_amd import _hip
    _hip.lib()
    return _hip


def dev(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda:0")


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def compare(out, ref, B):
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    for key in ("J", "mu", "delta", "xs", "us", "k", "K"):
        assert eq(out[key].cpu().numpy(), ref[key]), key
    tr = {key: out["trace"][key].cpu().numpy().T for key in out["trace"]}
    for b in range(B):
        it = int(ref["iters"][b])
        for key in ("J", "accepted", "alpha_idx", "mu", "delta", "qp_iter", "qp_backtrack", "n_clamped", "stopped_at"):
            assert eq(tr[key][b, :it], ref["trace"][key][b, :it]), (b, key)


def test_user_cartpole_equals_builtin_path_and_oracle(hip):
    from ilqr_amd import user_model as um
    T, B = 100, 200
    Mb, Mu = cartpole_as_user(T)
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=21, seed_u=22)
    ref = co.solve(Mu, co.Opts(max_iter=200), x0, us0, nthreads=8)
    M = um.UserModel(um.load("cartpole_user"), T, Mu.user_params, Mu.Q, Mu.R, Mu.Qt, Mu.x_goal, None, Mu.lower, Mu.upper)
    out = um.solve(M, hip.Opts(max_iter=200), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    compare(out, ref, B)
    Mh = hip.Model(T=T, Q=Mb.Q, R=Mb.R, Q_terminal=Mb.Qt, x_goal=Mb.x_goal, lower=Mb.lower, upper=Mb.upper)
    blt = hip.solve(Mh, hip.Opts(max_iter=200), dev(x0), dev(us0))
    for key in ("xs", "us", "J", "iters", "status"):
        assert eq(out[key].cpu().numpy(), blt[key].cpu().numpy()), key
    assert (ref["trace"]["n_clamped"] > 0).any()


def test_user_pendulum_equals_oracle(hip):
    from ilqr_amd import user_model as um
    T, B = 80, 333
    Mo = pendulum(T)
    x0, us0 = pendulum_batch(B, T, seed=31)
    ref = co.solve(Mo, co.Opts(max_iter=60), x0, us0, nthreads=8)
    M = um.UserModel(um.load("pendulum_user"), T, Mo.user_params, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, None, Mo.lower, Mo.upper)
    out = um.solve(M, hip.Opts(max_iter=60), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    compare(out, ref, B)
    assert (ref["trace"]["n_clamped"] > 0).any()


def test_forward_difference_user_model_equals_oracle(hip):
    """A model with no analytic Jacobian (forward differences, FiniteDifferenceDynamics' rule):
    the same f evaluations and the same divisions on both sides, so == holds here too."""
    from ilqr_amd import user_model as um
    T, B = 60, 150
    Mo = pendulum_fd(T)
    x0, us0 = pendulum_batch(B, T, seed=51)
    ref = co.solve(Mo, co.Opts(max_iter=40), x0, us0, nthreads=8)
    M = um.UserModel(um.load("pendulum_fd_user"), T, Mo.user_params, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, None, Mo.lower,
                     Mo.upper)
    out = um.solve(M, hip.Opts(max_iter=40), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    compare(out, ref, B)


@pytest.mark.parametrize("B,T,max_iter", [(150, 50, 40), (65, 9, 6), (1, 30, 25), (1500, 50, 100)])
def test_two_control_user_model_equals_oracle(hip, B, T, max_iter):
    """m = 2: the general path of the per-model solver (dense boxQP with its index compaction,
    dense Q-terms, cooperative line search) against the oracle's generic code on the same source;
    the largest case runs to the iteration limit and has searches that go past alpha_0."""
    from ilqr_amd import user_model as um
    Mo = rotor(T)
    x0, us0 = rotor_batch(B, T, seed=70 + B)
    ref = co.solve(Mo, co.Opts(max_iter=max_iter), x0, us0, nthreads=8)
    M = um.UserModel(um.load("planar_rotor_user"), T, Mo.user_params, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, None, Mo.lower,
                     Mo.upper)
    assert (M.n, M.m) == (7, 2)
    out = um.solve(M, hip.Opts(max_iter=max_iter), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    compare(out, ref, B)
    if B >= 100:
        assert (ref["trace"]["n_clamped"] > 0).any()
    if B >= 1000:
        assert (ref["trace"]["alpha_idx"] >= 1).any() and ref["iters"].max() == max_iter


def test_forward_difference_two_control_model_equals_oracle(hip):
    """ILQR_USER_FD with a control vector: the device evaluates the same forward differences as the oracle."""
    from ilqr_amd import user_model as um
    T, B = 30, 90
    Mo = rotor_fd(T)
    x0, us0 = rotor_batch(B, T, seed=121)
    ref = co.solve(Mo, co.Opts(max_iter=30), x0, us0, nthreads=8)
    M = um.UserModel(um.load("planar_rotor_fd_user"), T, Mo.user_params, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, None, Mo.lower,
                     Mo.upper)
    out = um.solve(M, hip.Opts(max_iter=30), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    compare(out, ref, B)


@pytest.mark.parametrize("B,T,max_iter", [(130, 30, 30), (700, 20, 40)])
def test_largest_user_model_equals_oracle(hip, B, T, max_iter):
    """8 states and 4 controls, the most the per-model solver takes (its matrices spill to scratch memory):
    every output and the per-iteration trace against the oracle's generic code on the same source."""
    from ilqr_amd import user_model as um
    Mo = chain(T)
    x0, us0 = chain_batch(B, T, seed=300 + B)
    ref = co.solve(Mo, co.Opts(max_iter=max_iter), x0, us0, nthreads=8)
    M = um.UserModel(um.load("mass_chain_user"), T, Mo.user_params, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, None, Mo.lower,
                     Mo.upper)
    assert (M.n, M.m) == (8, 4)
    out = um.solve(M, hip.Opts(max_iter=max_iter), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    compare(out, ref, B)
    assert (ref["trace"]["n_clamped"] > 0).any() and (ref["trace"]["n_clamped"] < 4).any()


def test_two_control_model_through_the_solver_api(hip):
    from ilqr_amd import QRCost, iLQR
    from ilqr_amd.user_model import UserDynamics
    from test_user_model_cpu import ROTOR_PARAMS
    T, B = 30, 11
    Mo = rotor(T)
    x0, us0 = rotor_batch(B, T, seed=91)
    ref = co.solve(Mo, co.Opts(max_iter=20), x0, us0, nthreads=8)
    dyn = UserDynamics("planar_rotor_user", ROTOR_PARAMS, max_control=3.0)
    assert dyn.control_dimension == 2 and dyn.state_dimension == 7
    s = iLQR(dyn, QRCost(Mo.Q, Mo.R, Q_terminal=Mo.Qt, x_goal=Mo.x_goal), T)
    xs, us = s.update_u(x0, us0, n_iterations=20)
    assert us.shape == (B, T, 2) and eq(xs, ref["xs"]) and eq(us, ref["us"]) and eq(s.K_, ref["K"])


@pytest.mark.parametrize("which", ["pendulum", "rotor"])
def test_user_models_track_a_path_with_pathqrcost(hip, which):
    """PathQRCost (ilqr/ilqr_cost.py:635-770: goals by time step, value without the 1/2) on both
    user-model paths, through the solver API, against the oracle."""
    from ilqr_amd import PathQRCost, iLQR
    from ilqr_amd.user_model import UserDynamics
    from test_user_model_cpu import ROTOR_PARAMS
    T, B = 40, 70
    if which == "pendulum":
        Mq = pendulum(T)
        x0, us0 = pendulum_batch(B, T, seed=101)
        th = np.linspace(0.0, 2.5, T + 1)
        x_path = np.stack([np.sin(th), np.cos(th), np.full(T + 1, 2.5 / (T * 0.05))], axis=1)
        u_path = 0.3 * np.cos(np.linspace(0, 3, T))[:, None]
        dyn = UserDynamics("pendulum_user", Mq.user_params, max_control=2.0)
    else:
        Mq = rotor(T)
        x0, us0 = rotor_batch(B, T, seed=102)
        s_ = np.linspace(0.0, 1.0, T + 1)
        x_path = np.stack([1.5 * s_, np.sin(3 * s_), np.zeros(T + 1), np.ones(T + 1), np.full(T + 1, 0.75),
                           np.cos(3 * s_), np.zeros(T + 1)], axis=1)
        u_path = 0.2 * np.stack([np.sin(np.linspace(0, 2, T)), np.cos(np.linspace(0, 2, T))], axis=1)
        dyn = UserDynamics("planar_rotor_user", ROTOR_PARAMS, max_control=3.0)
    Mo = co.user_model(Mq.user, Mq.user_params, T, Mq.Q, Mq.R, Mq.Qt, x_path, u_path, Mq.lower, Mq.upper,
                       cost_kind=co.COST_PATHQR)
    ref = co.solve(Mo, co.Opts(max_iter=30), x0, us0, nthreads=8)
    s = iLQR(dyn, PathQRCost(Mq.Q, Mq.R, x_path, u_path, Q_terminal=Mq.Qt), T)
    xs, us = s.update_u(x0, us0, n_iterations=30)
    assert eq(xs, ref["xs"]) and eq(us, ref["us"]) and eq(s.K_, ref["K"])
    assert np.array_equal(s.iterations_, ref["iters"])
    # the cost really is the path cost: twice the QRCost value of the same deviation at step 0
    c0 = co.user_model(Mq.user, Mq.user_params, T, Mq.Q, Mq.R, Mq.Qt, x_path[0], u_path[0], Mq.lower, Mq.upper)
    assert ref["J"].min() > 0 and not eq(ref["J"], co.solve(c0, co.Opts(max_iter=30), x0, us0, nthreads=8)["J"])


def test_user_model_through_the_solver_api(hip):
    from ilqr_amd import QRCost, iLQR
    from ilqr_amd.user_model import UserDynamics
    T, B = 50, 17
    Mo = pendulum(T)
    x0, us0 = pendulum_batch(B, T, seed=41)
    ref = co.solve(Mo, co.Opts(max_iter=30), x0, us0, nthreads=8)
    dyn = UserDynamics("pendulum_user", Mo.user_params, max_control=2.0)
    cost = QRCost(Mo.Q, Mo.R, Q_terminal=Mo.Qt, x_goal=Mo.x_goal)
    s = iLQR(dyn, cost, T)
    xs, us = s.update_u(x0, us0, n_iterations=30)
    assert xs.shape == (B, T + 1, 3) and eq(xs, ref["xs"]) and eq(us, ref["us"])
    assert np.array_equal(s.iterations_, ref["iters"])
    log = []
    t = iLQR(dyn, cost, T)
    xs1, us1 = t.update_u(x0[0], us0[0], n_iterations=30,
                          on_iteration=lambda it, xs_, us_, J, acc, conv: log.append((it, J, acc, conv)))
    it_n = int(ref["iters"][0])
    assert eq(us1, ref["us"][0]) and len(log) == it_n


@pytest.mark.parametrize("B,T,max_iter", [(1, 7, 12), (63, 1, 5), (65, 33, 0), (129, 20, 3)])
def test_user_model_ragged_batches_and_edge_sizes(hip, B, T, max_iter):
    """Partly filled wavefronts, a one-step horizon, zero iterations (initial rollout only)."""
    from ilqr_amd import user_model as um
    Mo = pendulum(T)
    x0, us0 = pendulum_batch(B, T, seed=60 + B)
    ref = co.solve(Mo, co.Opts(max_iter=max_iter), x0, us0, nthreads=4)
    M = um.UserModel(um.load("pendulum_user"), T, Mo.user_params, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, None, Mo.lower, Mo.upper)
    out = um.solve(M, hip.Opts(max_iter=max_iter), dev(x0), dev(us0), trace=max_iter > 0)
    torch.cuda.synchronize()
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    for key in ("J", "mu", "delta", "xs", "us", "k", "K"):
        assert eq(out[key].cpu().numpy(), ref[key]), key


def test_user_model_rejects_mismatched_dimensions(hip):
    from ilqr_amd import user_model as um
    lib = um.load("pendulum_user")
    M = um.UserModel(lib, 10, [0.05, 9.81, 0.2, 1.5], np.eye(3), [[1.0]], None, None, None, [-1.0], [1.0])
    M.c.n = 4
    with pytest.raises(hip.IlqrHipError):
        um.solve(M, hip.Opts(max_iter=2), dev(np.zeros((2, 3))), dev(np.zeros((2, 10, 1))))


def test_user_model_chunked_solve_with_compaction(hip):
    """The chunked solve (finished problems dropped between launches) with a per-model library
    as the solver: same results as its single launch."""
    from ilqr_amd import compaction, user_model as um
    T, B = 40, 500
    Mo = pendulum(T)
    x0, us0 = pendulum_batch(B, T, seed=111)
    M = um.UserModel(um.load("pendulum_user"), T, Mo.user_params, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, None, Mo.lower, Mo.upper)
    one = um.solve(M, hip.Opts(max_iter=50), dev(x0), dev(us0))
    chunked = compaction.solve_compacting(um.solve, M, {}, 50, dev(x0), dev(us0), chunk=6)
    torch.cuda.synchronize()
    for key in ("xs", "us", "J", "mu", "delta", "iters", "status", "k", "K"):
        assert torch.equal(one[key], chunked[key]), key
    it = one["iters"].cpu().numpy()
    assert (it < 50).any() and it.max() > 6
    # the multi-control kernel continues from carried state the same way
    T, B = 30, 300
    Mo = rotor(T)
    x0, us0 = rotor_batch(B, T, seed=131)
    M = um.UserModel(um.load("planar_rotor_user"), T, Mo.user_params, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, None, Mo.lower,
                     Mo.upper)
    one = um.solve(M, hip.Opts(max_iter=40), dev(x0), dev(us0))
    chunked = compaction.solve_compacting(um.solve, M, {}, 40, dev(x0), dev(us0), chunk=5)
    torch.cuda.synchronize()
    for key in ("xs", "us", "J", "mu", "delta", "iters", "status", "k", "K"):
        assert torch.equal(one[key], chunked[key]), key
    it = one["iters"].cpu().numpy()
    assert (it < 40).any() and it.max() > 5


def test_user_model_compaction_just_above_the_threshold(hip):
    """A user-model batch between 49152 and 65535 problems: when its running set falls to between half the
    batch and 32768 problems (here after 6 iterations: 32730 of 49852), the launch uses 32 problems per
    wavefront and needs a LARGER workspace than the full batch did; the workspace handed from launch to
    launch must be re-sized, and the results must be those of the single launch."""
    import ctypes
    from ilqr_amd import compaction, user_model as um
    T, B = 12, 49152 + 700
    Mo = pendulum(T)
    x0, us0 = pendulum_batch(B, T, seed=171)
    M = um.UserModel(um.load("pendulum_user"), T, Mo.user_params, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, None, Mo.lower, Mo.upper)

    def need(n):
        return int(M.lib.ilqr_user_workspace_bytes(ctypes.byref(M.c), ctypes.c_int64(n)))
    one = um.solve(M, hip.Opts(max_iter=24), dev(x0), dev(us0))
    it = one["iters"].cpu().numpy()
    running = [int((it > c).sum()) for c in range(0, 24, 2)]  # at the chunk boundaries
    assert any(n <= 32768 and need(n) > need(B) for n in running), running
    chunked = compaction.solve_compacting(um.solve, M, {}, 24, dev(x0), dev(us0), chunk=2)
    torch.cuda.synchronize()
    for key in ("xs", "us", "J", "mu", "delta", "iters", "status", "k", "K"):
        assert torch.equal(one[key], chunked[key]), key
