"""GPU parity of a user-written model with more than 8 states (tests/models/long_chain_user.h: 20 states, 5
controls) on the large-state path: the library is the MFMA solver compiled around the model source
(ilqr_amd.user_model.build_large_user_model, built by __graft_entry__.build()).  The reference's counterpart is a
Python Dynamics object evaluated once per time step (ilqr/ilqr_dynamics.py:13-91, :94-575); the check is the
oracle's generic solver running the SAME source compiled for the CPU.  Bar: ==."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import c_oracle as co  # noqa: E402

MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
PARAMS = [0.05, 4.0, 6.0, 0.3]


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ilqr_amd import _hip
    _hip.lib()
    return _hip


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def chain_model(T, cost_kind=co.COST_QR, x_goal=None, u_goal=None):
    lib = co.build_user_model(os.path.join(MODELS, "long_chain_user.h"), "long_chain_user")
    Q = np.diag([4.0] * 10 + [1.0] * 10)
    R = 0.2 * np.eye(5) + 0.02 * np.ones((5, 5))
    xg = np.concatenate([0.3 * np.arange(1, 11), np.zeros(10)]) if x_goal is None else x_goal
    return co.user_model(lib, PARAMS, T, Q, R, 30.0 * np.eye(20), xg, u_goal, [-1.5] * 5, [1.5] * 5, cost_kind=cost_kind)


def chain_batch(B, T, seed):
    r = np.random.RandomState(seed)
    x0 = np.concatenate([np.cumsum(r.uniform(0.0, 0.5, (B, 10)), axis=1), r.uniform(-0.5, 0.5, (B, 10))], axis=1)
    return x0, r.uniform(-1.0, 1.0, (B, T, 5))


def compare(out, ref, B):
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    for key in ("J", "mu", "delta", "xs", "us", "k", "K"):
        assert eq(out[key].cpu().numpy(), ref[key]), key
    if out.get("trace") is not None:
        tr = {key: out["trace"][key].cpu().numpy().T for key in out["trace"]}
        for b in range(B):
            it = int(ref["iters"][b])
            for key in ("J", "accepted", "alpha_idx", "mu", "delta", "qp_iter", "qp_backtrack", "n_clamped", "stopped_at"):
                assert eq(tr[key][b, :it], ref["trace"][key][b, :it]), (b, key)


def test_twenty_state_user_model_equals_oracle(hip):
    from ilqr_amd import user_model as um
    T, B, n_it = 40, 70, 30
    Mo = chain_model(T)
    x0, us0 = chain_batch(B, T, 91)
    ref = co.solve(Mo, co.Opts(max_iter=n_it), x0, us0, nthreads=8)
    lib = um.load_large("long_chain_user")
    assert um.large_user_dims(lib) == (20, 5, 4)
    M = um.LargeUserModel(lib, T, PARAMS, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, Mo.u_goal, Mo.lower, Mo.upper)
    out = hip.large_solve(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    compare(out, ref, B)
    assert (ref["trace"]["n_clamped"] > 0).any() and ref["iters"].max() > 5
    # stage level: rollout, one backward sweep (result codes, free sets, counters)
    xs_ref, J_ref = co.rollout_cost(Mo, x0, us0)
    xs, J = hip.large_rollout_cost(M, dev(x0), dev(us0))
    assert eq(xs.cpu().numpy(), xs_ref) and eq(J.cpu().numpy(), J_ref)
    bw_ref = co.backward(Mo, co.Opts(), xs_ref, us0, np.ones(B))
    bw = hip.large_backward(M, hip.Opts(), dev(xs_ref), dev(us0), dev(np.ones(B)))
    assert eq(bw["k"].cpu().numpy(), bw_ref["k"]) and eq(bw["K"].cpu().numpy(), bw_ref["K"])
    assert np.array_equal(bw["qp_result"].cpu().numpy(), bw_ref["qp_result"])
    assert np.array_equal(bw["free"].cpu().numpy(), bw_ref["free"])
    # the derivative stream belongs to the built-in model family
    with pytest.raises(hip.IlqrHipError):
        hip.large_deriv(M, dev(xs_ref), dev(us0))


def test_twenty_state_user_model_through_the_solver_api(hip):
    from ilqr_amd import PathQRCost, QRCost, iLQR
    from ilqr_amd.user_model import LargeUserDynamics
    T, B, n_it = 25, 12, 15
    Mo = chain_model(T)
    x0, us0 = chain_batch(B, T, 92)
    ref = co.solve(Mo, co.Opts(max_iter=n_it), x0, us0, nthreads=8)
    dyn = LargeUserDynamics("long_chain_user", PARAMS, 1.5)
    assert (dyn.state_dimension, dyn.control_dimension) == (20, 5)
    s = iLQR(dyn, QRCost(Mo.Q, Mo.R, Q_terminal=Mo.Qt, x_goal=Mo.x_goal), T)
    xs, us = s.update_u(x0, us0, n_iterations=n_it)
    assert xs.shape == (B, T + 1, 20) and eq(xs, ref["xs"]) and eq(us, ref["us"]) and eq(s.K_, ref["K"])
    log = []
    t = iLQR(dyn, QRCost(Mo.Q, Mo.R, Q_terminal=Mo.Qt, x_goal=Mo.x_goal), T)
    xs1, us1 = t.update_u(x0[0], us0[0], n_iterations=n_it, on_iteration=lambda it, *a: log.append(a[2]))
    it0 = int(ref["iters"][0])
    assert eq(us1, ref["us"][0]) and eq(log, ref["trace"]["J"][0, :it0])
    # a path per problem
    r = np.random.RandomState(93)
    xp = np.tile(np.asarray(Mo.x_goal)[None, None], (B, T + 1, 1))
    xp[:, :, :10] += np.cumsum(r.uniform(-0.02, 0.02, (B, T + 1, 1)), axis=1)
    up = r.uniform(-0.1, 0.1, (B, T, 5))
    outs = [co.solve(chain_model(T, co.COST_PATHQR, xp[b], up[b]), co.Opts(max_iter=n_it), x0[b:b + 1], us0[b:b + 1],
                     trace=False) for b in range(B)]
    sp = iLQR(dyn, PathQRCost(Mo.Q, Mo.R, xp, up, Q_terminal=Mo.Qt), T)
    xs2, us2 = sp.update_u(x0, us0, n_iterations=n_it)
    assert eq(us2, np.concatenate([o["us"] for o in outs])) and eq(sp.J_, np.concatenate([o["J"] for o in outs]))
