"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bar: every floating-point output equal to the oracle's (deterministic math mode)
as VALUES, compared with ==, i.e. bit-exact except that -0.0 == +0.0; every
integer output (iteration counts, boxQP result codes, active sets, line-search
indices) identical.  The fixtures recorded from the reference's own solver are
matched to 1e-9 relative on costs with identical discrete decisions.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import c_oracle as co  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ilqr_amd import _hip
    _hip.lib()
    return _hip


DEV = "cuda:0"


def dev(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def hip_model(hip, Mo):
    p = Mo.params
    return hip.Model(Mo.T, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, Mo.u_goal, Mo.lower, Mo.upper, dt=p["dt"], mc=p["mc"],
                     mp=p["mp"], l=p["l"], g=p["g"])


def eq(a, b):
    """Value equality incl. NaN positions."""
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def test_device_math_bit_exact(hip):
    r = np.random.RandomState(0)
    a = np.concatenate([r.uniform(-10, 10, 300000), r.uniform(-1e5, 1e5, 100000), r.uniform(-4, 4, 100000),
                        r.uniform(-1e-3, 1e-3, 20000), 10.0 ** r.uniform(-300, 300, 50000),
                        [0.0, -0.0, np.inf, -np.inf, np.nan, 2.0 ** 30, 2.0 ** 30 + 1, 1e300]])
    b = np.concatenate([r.uniform(-10, 10, a.size - 8), [0.0, 1.0, np.inf, -np.inf, 1.0, np.nan, -0.0, 1e-300]])
    s, c, t, sq, dv = [o.cpu().numpy() for o in hip.detmath(dev(a), dev(b))]
    so, cco, to = co.detmath(a, b)
    assert eq(s, so) and eq(c, cco) and eq(t, to)
    assert np.array_equal(np.signbit(t), np.signbit(to))
    # IEEE sqrt and divide on the device are correctly rounded (the oracle relies on it)
    with np.errstate(all="ignore"):
        assert eq(sq, np.sqrt(np.abs(a)))
        assert eq(dv, a / b)


@pytest.mark.parametrize("B,T", [(1, 100), (300, 100), (64, 15)])
def test_rollout_cost_bit_exact(hip, B, T):
    Mo = co.cartpole_model(T=T)
    x0, us = co.cartpole_batch(B, T=T, seed_x=3, seed_u=4)
    us *= 4.0
    xs_o, J_o = co.rollout_cost(Mo, x0, us)
    M = hip_model(hip, Mo)
    xs_bm, J = hip.rollout_cost_bm(M, B, hip.pack_bm(dev(x0)), hip.pack_bm(dev(us)))
    xs = hip.unpack_bm(xs_bm, B, (T + 1, 5)).cpu().numpy()
    assert eq(xs, xs_o) and eq(J.cpu().numpy(), J_o)


def _trajectories(B, T, scale_u, seed):
    Mo = co.cartpole_model(T=T)
    x0, us = co.cartpole_batch(B, T=T, seed_x=seed, seed_u=seed + 1)
    us *= scale_u
    xs, _ = co.rollout_cost(Mo, x0, us)
    return Mo, xs, us


@pytest.mark.parametrize("B,T,scale_u,mu", [(200, 100, 1.0, 1.0), (130, 100, 6.0, 0.0), (64, 15, 3.0, 1e3)])
def test_backward_pass_bit_exact(hip, B, T, scale_u, mu):
    Mo, xs, us = _trajectories(B, T, scale_u, 7)
    mus = np.full(B, mu)
    mus[::3] *= 0.5
    ref = co.backward(Mo, co.Opts(), xs, us, mus)
    M = hip_model(hip, Mo)
    out = hip.backward_bm(M, hip.Opts(), B, hip.pack_bm(dev(xs)), hip.pack_bm(dev(us)), dev(mus))
    k = hip.unpack_bm(out["k"], B, (T, 1)).cpu().numpy()
    K = hip.unpack_bm(out["K"], B, (T, 1, 5)).cpu().numpy()
    assert eq(k, ref["k"]) and eq(K, ref["K"])
    dJ = out["dJ"].cpu().numpy()[:, :B]
    info = out["info"].cpu().numpy()[:, :B]
    assert eq(dJ[0], ref["dJ1"]) and eq(dJ[1], ref["dJ2"])
    assert np.array_equal(info[0], ref["stopped_at"])
    assert np.array_equal(info[1], ref["qp_iter"])
    assert np.array_equal(info[2], ref["qp_backtrack"])
    assert np.array_equal(info[3], ref["n_clamped"])
    assert np.array_equal(out["qp_result"].cpu().numpy()[:, :B].T, ref["qp_result"])
    assert np.array_equal(out["free"].cpu().numpy()[:, :B].T.reshape(B, T, 1), ref["free"])
    # the workload really exercises the clamped / line-search-exhausted exits
    if scale_u >= 3.0:
        assert (ref["n_clamped"] > 0).any()


@pytest.mark.parametrize("cost", ["qr", "path"])
def test_staged_backward_from_derivative_streams(hip, golden_dir, cost):
    """Three-stage form: the derivative kernel writes f_x, f_u, l_x, ... as HBM streams and the
    backward sweep reads them back.  Same gains, same boxQP behaviour, bit for bit, as the
    oracle (and hence as the fused sweep)."""
    if cost == "qr":
        B, T = 150, 100
        Mo, xs, us = _trajectories(B, T, 4.0, 15)
        M = hip_model(hip, Mo)
    else:
        g, Mo, M = pathqr_models(hip, golden_dir)
        B, T = 90, Mo.T
        x0 = co.cartpole_batch(B, T=T, seed_x=73, seed_u=74)[0]
        us = np.random.RandomState(6).uniform(-3, 3, (B, T, 1))
        xs, _ = co.rollout_cost(Mo, x0, us)
    mus = np.full(B, 0.5)
    ref = co.backward(Mo, co.Opts(), xs, us, mus)
    xs_bm, us_bm = hip.pack_bm(dev(xs)), hip.pack_bm(dev(us))
    d = hip.deriv_bm(M, B, xs_bm, us_bm)
    out = hip.backward_bm(M, hip.Opts(), B, None, us_bm, dev(mus), deriv=d)
    fused = hip.backward_bm(M, hip.Opts(), B, xs_bm, us_bm, dev(mus))
    for o in (out, fused):
        assert eq(hip.unpack_bm(o["k"], B, (T, 1)).cpu().numpy(), ref["k"])
        assert eq(hip.unpack_bm(o["K"], B, (T, 1, 5)).cpu().numpy(), ref["K"])
        assert eq(o["dJ"].cpu().numpy()[0, :B], ref["dJ1"]) and eq(o["dJ"].cpu().numpy()[1, :B], ref["dJ2"])
        assert np.array_equal(o["info"].cpu().numpy()[2, :B], ref["qp_backtrack"])
        assert np.array_equal(o["qp_result"].cpu().numpy()[:, :B].T, ref["qp_result"])
    assert (ref["n_clamped"] > 0).any()


def test_backward_pass_not_pd_break(hip):
    """A large negative terminal weight makes Q_uu lose positive definiteness: the
    reference breaks out of the sweep and leaves the earlier gains zero (ilqr_core.py:137-140)."""
    T, B = 30, 70
    Qt = -4000.0 * np.eye(5)
    Mo = co.cartpole_model(T=T)
    Mo = co.Model(co.MODEL_CARTPOLE, 5, 1, T, co.COST_QR, co.MATH_DET, True, Q=Mo.Q, R=Mo.R, Q_terminal=Qt,
                  x_goal=Mo.x_goal, lower=Mo.lower, upper=Mo.upper)
    x0, us = co.cartpole_batch(B, T=T, seed_x=9, seed_u=10)
    xs, _ = co.rollout_cost(Mo, x0, us)
    ref = co.backward(Mo, co.Opts(), xs, us, 1.0)
    assert (ref["stopped_at"] >= 0).any()
    M = hip_model(hip, Mo)
    out = hip.backward_bm(M, hip.Opts(), B, hip.pack_bm(dev(xs)), hip.pack_bm(dev(us)), dev(np.full(B, 1.0)))
    assert eq(hip.unpack_bm(out["k"], B, (T, 1)).cpu().numpy(), ref["k"])
    assert eq(hip.unpack_bm(out["K"], B, (T, 1, 5)).cpu().numpy(), ref["K"])
    assert np.array_equal(out["info"].cpu().numpy()[0, :B], ref["stopped_at"])
    assert np.array_equal(out["qp_result"].cpu().numpy()[:, :B].T, ref["qp_result"])


def test_forward_line_search_bit_exact(hip):
    B, T = 150, 100
    Mo, xs, us = _trajectories(B, T, 2.0, 11)
    bw = co.backward(Mo, co.Opts(), xs, us, 1.0)
    alphas = co.line_search_alphas(30)
    xs_n, us_n, J_n = co.forward(Mo, alphas, xs, us, bw["k"], bw["K"])
    M = hip_model(hip, Mo)
    J, xs_new, us_new = hip.forward_ls_bm(M, B, alphas, hip.pack_bm(dev(xs)), hip.pack_bm(dev(us)),
                                          hip.pack_bm(dev(bw["k"])), hip.pack_bm(dev(bw["K"])), want_traj=True)
    assert eq(J.cpu().numpy()[:, :B].T, J_n)
    xs_new = xs_new.cpu().numpy()[:, :, :B].transpose(2, 0, 1).reshape(B, 30, T + 1, 5)
    us_new = us_new.cpu().numpy()[:, :, :B].transpose(2, 0, 1).reshape(B, 30, T, 1)
    assert eq(xs_new, xs_n) and eq(us_new, us_n)


def test_derivative_streams_bit_exact(hip):
    B, T = 70, 100
    Mo, xs, us = _trajectories(B, T, 2.0, 13)
    M = hip_model(hip, Mo)
    out = hip.deriv_bm(M, B, hip.pack_bm(dev(xs)), hip.pack_bm(dev(us)))
    shapes = dict(fx=(T, 5, 5), fu=(T, 5, 1), lx=(T, 5), lu=(T, 1), lxx=(T, 5, 5), luu=(T, 1, 1), lux=(T, 1, 5),
                  lxT=(5,), lxxT=(5, 5))
    got = {k: hip.unpack_bm(out[k], B, shapes[k]).cpu().numpy() for k in shapes}
    for b in range(0, B, 7):
        ref = co.derivatives(Mo, xs[b], us[b])
        for k in shapes:
            assert eq(got[k][b], ref[k]), (b, k)


def _check_solve(hip, Mo, x0, us0, max_iter, delta=None, k=None, K=None):
    B = x0.shape[0]
    ref = co.solve(Mo, co.Opts(max_iter=max_iter), x0, us0, delta=delta, k=k, K=K, nthreads=8)
    M = hip_model(hip, Mo)
    out = hip.solve(M, hip.Opts(max_iter=max_iter), dev(x0), dev(us0),
                    delta=None if delta is None else dev(delta), k=None if k is None else dev(k),
                    K=None if K is None else dev(K), trace=True)
    torch.cuda.synchronize()
    g = {key: out[key].cpu().numpy() for key in ("xs", "us", "J", "mu", "delta", "iters", "status", "k", "K")}
    assert np.array_equal(g["iters"], ref["iters"])
    assert np.array_equal(g["status"], ref["status"])
    for key in ("J", "mu", "delta", "xs", "us", "k", "K"):
        assert eq(g[key], ref[key]), key
    tr = {key: out["trace"][key].cpu().numpy().T for key in out["trace"]}
    for b in range(B):
        it = ref["iters"][b]
        for key in ("J", "mu", "delta"):
            assert eq(tr[key][b, :it], ref["trace"][key][b, :it]), (b, key)
        for key in ("accepted", "alpha_idx", "qp_iter", "qp_backtrack", "n_clamped", "stopped_at"):
            assert np.array_equal(tr[key][b, :it], ref["trace"][key][b, :it]), (b, key)
        assert (tr["accepted"][b, it:] == -1).all()
    return ref, g


def test_solve_bit_exact_swingup_batch(hip):
    """C1-style batch (random swing-up states, bounds +-5 active)."""
    B = 192
    x0, us0 = co.cartpole_batch(B)
    ref, _ = _check_solve(hip, co.cartpole_model(), x0, us0, 200)
    assert (ref["status"] == 1).any() and (ref["status"] == 0).any()
    assert (ref["trace"]["n_clamped"] > 0).any()


def test_solve_bit_exact_ragged_and_warm_state(hip):
    """B not a multiple of 64, short horizon, and a second call that reuses the
    solver state (delta_, k_, K_) of the first, as the MPC driver does."""
    B, T = 77, 15
    Mo = co.cartpole_model(T=T, max_control=20.0, l=0.5, mpc=True)
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=5, seed_u=6)
    ref, g = _check_solve(hip, Mo, x0, us0, 25)
    us1 = np.concatenate([g["us"][:, 1:], g["us"][:, -1:]], axis=1)
    _check_solve(hip, Mo, g["xs"][:, 1], us1, 10, delta=g["delta"], k=g["k"], K=g["K"])


def test_solve_matches_reference_run_fixtures(hip, golden_dir):
    """Against numbers recorded from the reference's own update_u."""
    g = np.load(golden_dir + "/ref_cartpole_solve.npz")
    Mo = co.cartpole_model()
    M = hip_model(hip, Mo)
    for i in range(int(g["n_problems"])):
        p = "sympy_p%d_" % i
        mi = int(g[p + "max_iter"])
        out = hip.solve(M, hip.Opts(max_iter=mi), dev(g[p + "x0"][None]), dev(g[p + "us_init"][None]), trace=True)
        it = int(out["iters"][0])
        ref_J = g[p + "it_J"]
        assert it == len(ref_J)
        tr = {key: out["trace"][key].cpu().numpy()[:it, 0] for key in out["trace"]}
        np.testing.assert_allclose(tr["J"], ref_J, rtol=1e-9)
        assert np.array_equal(tr["accepted"].astype(bool), g[p + "it_accepted"])
        assert np.array_equal(np.minimum(tr["alpha_idx"] + 1, 30), g[p + "it_n_forward"])
        assert np.array_equal(tr["mu"], g[p + "it_mu"]) and np.array_equal(tr["delta"], g[p + "it_delta"])
        assert bool(out["status"][0] == 1) == bool(g[p + "it_converged"][-1])
        np.testing.assert_allclose(out["us"][0].cpu().numpy(), g[p + "us"], rtol=0, atol=1e-8)
        # the active sets of the reference's own boxQP calls (ilqr_core.py:156), per iteration: steps with a
        # clamped control, and the step of a non-PD break
        reached = g[p + "it_qp_result"] != -100
        assert np.array_equal(tr["n_clamped"], (reached & ~g[p + "it_free"][:, :, 0]).sum(axis=1))
        assert np.array_equal(tr["stopped_at"],
                              np.where(reached.all(axis=1), -1, reached.shape[1] - 1 - reached[:, ::-1].sum(axis=1)))


def test_boxqp_batched(hip, golden_dir):
    g = np.load(golden_dir + "/ref_boxqp.npz")
    for i in range(int(g["n_cases"])):
        p = "q%d_" % i
        m = g[p + "g"].shape[0]
        x0 = g[p + "x0"][None] if bool(g[p + "has_x0"]) else None
        O = hip.Opts(qp_step_dec=float(g[p + "step_dec"]))
        ref = co.boxqp(g[p + "H"], g[p + "g"], g[p + "lower"], g[p + "upper"],
                       None if x0 is None else x0, co.Opts(qp_step_dec=float(g[p + "step_dec"])))
        out = hip.boxqp_batched(dev(g[p + "H"][None]), dev(g[p + "g"][None]), dev(g[p + "lower"][None]),
                                dev(g[p + "upper"][None]), None if x0 is None else dev(x0), O)
        assert int(out["result"][0]) == int(g[p + "result"]) == ref["result"][0], i
        assert np.array_equal(out["free"][0].cpu().numpy().astype(bool), g[p + "free"])
        assert eq(out["x"][0].cpu().numpy(), ref["x"][0])
        d = int(out["hff_dim"][0])
        assert d == ref["hff_dim"][0]
        assert eq(np.tril(out["Hff"][0].cpu().numpy()[:d, :d]), np.tril(ref["Hff"][0][:d, :d]))
        assert np.array_equal(out["counters"][0].cpu().numpy(),
                              [ref["n_iter"][0], ref["n_factor"][0], ref["n_backtrack"][0]])
        np.testing.assert_allclose(out["x"][0].cpu().numpy(), g[p + "x"], rtol=1e-9, atol=1e-12)


def test_boxqp_batched_random_bit_exact(hip):
    r = np.random.RandomState(3)
    for m in (1, 2, 5, 8):
        B = 500
        A = r.randn(B, m, m)
        H = np.einsum("bij,bkj->bik", A, A) + 0.05 * np.eye(m)
        g_ = r.randn(B, m) * 3
        lo, up = -r.uniform(0.05, 2, (B, m)), r.uniform(0.05, 2, (B, m))
        x0 = np.where(r.uniform(size=(B, m)) < 0.3, up, r.randn(B, m))
        ref = co.boxqp(H, g_, lo, up, x0, co.Opts())
        out = hip.boxqp_batched(dev(H), dev(g_), dev(lo), dev(up), dev(x0), hip.Opts())
        assert np.array_equal(out["result"].cpu().numpy(), ref["result"])
        assert eq(out["x"].cpu().numpy(), ref["x"])
        assert np.array_equal(out["free"].cpu().numpy().astype(bool), ref["free"])
        assert np.array_equal(out["counters"].cpu().numpy()[:, 2], ref["n_backtrack"])


@pytest.mark.parametrize("step_dec,armijo,min_step", [(0.9, 0.1, 1e-22), (0.6, 0.1, 1e-22), (0.5, 0.4, 1e-12),
                                                      (0.97, 0.01, 1e-8), (0.9, 0.0, 1e-22), (0.9, 1.5, 1e-6)])
def test_boxqp1_shortcuts_over_magnitudes(hip, step_dec, armijo, min_step):
    """The m = 1 routine skips Armijo trips whose outcome it can prove (jump to the trip
    where the test flips, early exit from a search that cannot succeed).  Scalar QPs
    over 30 decades of curvature and gradient, warm starts near and far from the
    optimum and from the bounds: result, solution bits and the backtrack COUNT must be
    those of the literal loop, for several option sets (including Armijo constants
    outside (0, 1), for which no shortcut applies)."""
    r = np.random.RandomState(17)
    B = 200000
    H = 10.0 ** r.uniform(-6, 11, B)
    g_ = r.choice([-1.0, 1.0], B) * 10.0 ** r.uniform(-12, 6, B)
    lo = -10.0 ** r.uniform(-3, 1, B)
    up = 10.0 ** r.uniform(-3, 1, B)
    opt = np.clip(-g_ / H, lo, up)
    kind = r.randint(0, 5, B)
    x0 = np.select([kind == 0, kind == 1, kind == 2, kind == 3],
                   [g_ / H,                                        # the backward pass's warm start (ilqr_core.py:145)
                    opt * (1 + 10.0 ** r.uniform(-16, -1, B)),     # next to the optimum
                    opt + r.choice([-1, 1], B) * 10.0 ** r.uniform(-12, 0, B),
                    np.where(r.uniform(size=B) < 0.5, lo, up)], r.randn(B) * 3)
    kw = dict(qp_step_dec=step_dec, qp_armijo=armijo, qp_min_step=min_step)
    sh = (B, 1)
    ref = co.boxqp(H.reshape(B, 1, 1), g_.reshape(sh), lo.reshape(sh), up.reshape(sh), x0.reshape(sh), co.Opts(**kw))
    out = hip.boxqp_batched(dev(H.reshape(B, 1, 1)), dev(g_.reshape(sh)), dev(lo.reshape(sh)), dev(up.reshape(sh)),
                            dev(x0.reshape(sh)), hip.Opts(**kw))
    assert np.array_equal(out["result"].cpu().numpy(), ref["result"])
    assert eq(out["x"].cpu().numpy(), ref["x"])
    cnt = out["counters"].cpu().numpy()
    assert np.array_equal(cnt[:, 0], ref["n_iter"]) and np.array_equal(cnt[:, 2], ref["n_backtrack"])
    assert len(set(ref["result"].tolist())) >= 3


def pathqr_models(hip, golden_dir, T=None):
    g = np.load(golden_dir + "/ref_pathqr_solve.npz")
    N = int(g["N"]) if T is None else T
    xp, up = g["x_path"][:N + 1], g["u_path"][:N]
    Mo = co.Model(co.MODEL_CARTPOLE, 5, 1, N, co.COST_PATHQR, co.MATH_DET, True, dt=0.05, Q=g["Q"], R=g["R"],
                  Q_terminal=g["Qt"], x_goal=xp, u_goal=up, lower=[-5.0], upper=[5.0])
    M = hip.Model(N, g["Q"], g["R"], g["Qt"], None, None, [-5.0], [5.0], dt=0.05, x_path=dev(xp), u_path=dev(up))
    return g, Mo, M


def test_pathqr_stages_bit_exact(hip, golden_dir):
    """PathQRCost (time-indexed goals, value without the 1/2, non-symmetric Q) through every
    stage kernel against the oracle."""
    g, Mo, M = pathqr_models(hip, golden_dir)
    N, B = Mo.T, 150
    r = np.random.RandomState(5)
    x0 = co.cartpole_batch(B, T=N, seed_x=71, seed_u=72)[0]
    us = r.uniform(-3, 3, (B, N, 1))
    xs_ref, J_ref = co.rollout_cost(Mo, x0, us)
    xs_bm, J = hip.rollout_cost_bm(M, B, hip.pack_bm(dev(x0)), hip.pack_bm(dev(us)))
    assert eq(hip.unpack_bm(xs_bm, B, (N + 1, 5)).cpu().numpy(), xs_ref) and eq(J.cpu().numpy()[:B], J_ref)
    us_bm = hip.pack_bm(dev(us))
    d = hip.deriv_bm(M, B, xs_bm, us_bm)
    for b in (0, 77, B - 1):
        dr = co.derivatives(Mo, xs_ref[b], us[b])
        assert eq(hip.unpack_bm(d["lx"], B, (N, 5)).cpu().numpy()[b], dr["lx"])
        assert eq(hip.unpack_bm(d["lu"], B, (N, 1)).cpu().numpy()[b], dr["lu"])
        assert eq(hip.unpack_bm(d["lxT"], B, (5,)).cpu().numpy()[b], dr["lxT"])
        assert eq(hip.unpack_bm(d["lxx"], B, (N, 5, 5)).cpu().numpy()[b], dr["lxx"])
    mu = np.full(B, 0.5)
    bw_ref = co.backward(Mo, co.Opts(), xs_ref, us, mu)
    bw = hip.backward_bm(M, hip.Opts(), B, xs_bm, us_bm, dev(mu))
    k, K = hip.unpack_bm(bw["k"], B, (N, 1)).cpu().numpy(), hip.unpack_bm(bw["K"], B, (N, 1, 5)).cpu().numpy()
    assert eq(k, bw_ref["k"]) and eq(K, bw_ref["K"])
    alphas = co.line_search_alphas(6)
    _, _, J_ref_a = co.forward(Mo, alphas, xs_ref, us, bw_ref["k"], bw_ref["K"])
    J_a, _, _ = hip.forward_ls_bm(M, B, alphas, xs_bm, us_bm, bw["k"], bw["K"])
    assert eq(J_a.cpu().numpy()[:, :B].T, J_ref_a)


def test_pathqr_solve_bit_exact_and_matches_reference_run(hip, golden_dir):
    g, Mo, M = pathqr_models(hip, golden_dir)
    N = Mo.T
    # the three recorded reference runs: identical decisions, costs within 1e-9 relative
    for i in range(int(g["n_problems"])):
        p = "p%d_" % i
        out = hip.solve(M, hip.Opts(max_iter=int(g[p + "max_iter"])), dev(g[p + "x0"][None]), dev(g[p + "us_init"][None]),
                        trace=True)
        it = int(out["iters"][0])
        assert it == len(g[p + "it_J"])
        np.testing.assert_allclose(out["trace"]["J"][:it, 0].cpu().numpy(), g[p + "it_J"], rtol=1e-9)
        assert np.array_equal(out["trace"]["accepted"][:it, 0].cpu().numpy().astype(bool), g[p + "it_accepted"])
        np.testing.assert_allclose(out["us"][0].cpu().numpy(), g[p + "us"], rtol=0, atol=1e-8)
    # a batch against the oracle, bit for bit
    B = 200
    x0, us0 = co.cartpole_batch(B, T=N, seed_x=81, seed_u=82)
    ref = co.solve(Mo, co.Opts(max_iter=60), x0, us0, nthreads=8, trace=False)
    out = hip.solve(M, hip.Opts(max_iter=60), dev(x0), dev(us0))
    for key in ("xs", "us", "J", "k", "K", "mu", "delta"):
        assert eq(out[key].cpu().numpy(), ref[key]), key
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])


def test_plant_step_bit_exact(hip):
    r = np.random.RandomState(8)
    st = r.uniform(-3, 3, (1000, 4))
    u = r.uniform(-20, 20, 1000)
    assert eq(hip.plant_step(dev(st), dev(u)).cpu().numpy(), co.plant_step(st, u))


@pytest.mark.parametrize("B", [16384, 20480, 32768, 40000])
def test_full_size_batch_properties(hip, B):
    """C1 at its full size (16384 problems: three wavefronts per group of 64) and at the sizes that take the other
    launch shapes (two wavefronts per group up to 32768, one beyond): properties that need no oracle run, and every
    problem of every launch compared with the oracle."""
    x0, us0 = co.cartpole_batch(B)
    Mo = co.cartpole_model()
    M = hip_model(hip, Mo)
    out = hip.solve(M, hip.Opts(), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    iters, status = out["iters"].cpu().numpy(), out["status"].cpu().numpy()
    J = out["J"].cpu().numpy()
    trJ, acc = out["trace"]["J"].cpu().numpy(), out["trace"]["accepted"].cpu().numpy()
    assert np.isfinite(J).all() and iters.min() >= 1 and iters.max() <= 200
    assert set(np.unique(status)) <= {0, 1, 2}
    # the cost never increases from one iteration to the next and the last entry is the result
    for b in range(0, B, 97):
        h = trJ[:iters[b], b]
        assert (np.diff(h) <= 0).all() and h[-1] == J[b]
        assert (acc[iters[b]:, b] == -1).all()
    # the returned states are exactly the rollout of the returned controls, and J is their cost
    xs_bm, J2 = hip.rollout_cost_bm(M, B, hip.pack_bm(out["xs"][:, 0].contiguous()), hip.pack_bm(out["us"]))
    assert eq(hip.unpack_bm(xs_bm, B, (101, 5)).cpu().numpy(), out["xs"].cpu().numpy())
    assert eq(J2.cpu().numpy(), J)
    # EVERY problem of the launch against the oracle, bit for bit (the oracle solves them on the host's cores)
    ref = co.solve(Mo, co.Opts(), x0, us0, nthreads=16, trace=False)
    assert np.array_equal(ref["iters"], iters) and np.array_equal(ref["status"], status)
    for key in ("J", "us", "xs", "k", "K", "mu", "delta"):
        assert eq(out[key].cpu().numpy(), ref[key]), key
    # sharding: a slice solved alone gives the same bits as inside the big batch
    lo, hi = 4096 + 17, 4096 + 17 + 1000
    part = hip.solve(M, hip.Opts(), dev(x0[lo:hi]), dev(us0[lo:hi]))
    assert eq(part["us"].cpu().numpy(), out["us"].cpu().numpy()[lo:hi])
    assert np.array_equal(part["iters"].cpu().numpy(), iters[lo:hi])


@pytest.mark.parametrize("B", [16384, 32768])
def test_every_problem_of_the_benchmark_batch_against_the_oracle(hip, B):
    """The batch bench.py times (bench.make_batch(B, 0)) in the launch shapes it is timed in -- 16384 problems:
    three wavefronts per group of 64 problems on every CU; 32768: two wavefronts per group -- with every
    problem's costs, controls, states, gains, regularisation state, iteration count and status == the oracle's."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    x0, us0 = bench.make_batch(B, 0)
    Mo = co.cartpole_model()
    ref = co.solve(Mo, co.Opts(), x0, us0, nthreads=16, trace=False)
    out = hip.solve(hip_model(hip, Mo), hip.Opts(), dev(x0), dev(us0))
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    for key in ("J", "us", "xs", "k", "K", "mu", "delta"):
        assert eq(out[key].cpu().numpy(), ref[key]), key


@pytest.mark.parametrize("seed", [11, 12])
def test_every_problem_of_a_mid_size_batch_against_the_oracle(hip, seed):
    """4096 full-length solves (up to 200 iterations, limits active), every problem compared with the
    oracle: the rare paths (boxQP fallbacks, searches past alpha_0, mu_max exits) occur somewhere in a batch
    of this size, in wavefronts that are half filled and share workgroups."""
    B = 4096
    x0, us0 = co.cartpole_batch(B, seed_x=seed, seed_u=100 + seed)
    Mo = co.cartpole_model()
    ref = co.solve(Mo, co.Opts(), x0, us0, nthreads=16, trace=False)
    out = hip.solve(hip_model(hip, Mo), hip.Opts(), dev(x0), dev(us0))
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    for key in ("J", "us", "xs", "k", "K", "mu", "delta"):
        assert eq(out[key].cpu().numpy(), ref[key]), key


def test_one_gpu_share_of_the_million_problem_configuration(hip):
    """131072 problems = one GPU's slice of the 8-GPU, 2^20-problem configuration (4-wave
    workgroups, two rounds of waves per SIMD): the first 16384 are the benchmark batch, so they
    must come out exactly as in the 16384-problem launch; a sample of the rest is checked
    against the oracle; returned states must be the rollout of the returned controls."""
    B = 131072
    x0, us0 = co.cartpole_batch(B)
    Mo = co.cartpole_model()
    M = hip_model(hip, Mo)
    out = hip.solve(M, hip.Opts(), dev(x0), dev(us0))
    small = hip.solve(M, hip.Opts(), dev(x0[:16384]), dev(us0[:16384]))
    torch.cuda.synchronize()
    iters, J = out["iters"].cpu().numpy(), out["J"].cpu().numpy()
    assert np.isfinite(J).all() and iters.min() >= 1 and iters.max() <= 200
    assert eq(small["us"].cpu().numpy(), out["us"][:16384].cpu().numpy())
    assert np.array_equal(small["iters"].cpu().numpy(), iters[:16384])
    idx = np.arange(16384 + 5, B, 4099)
    ref = co.solve(Mo, co.Opts(), x0[idx], us0[idx], nthreads=8, trace=False)
    assert np.array_equal(ref["iters"], iters[idx]) and eq(ref["J"], J[idx])
    assert eq(ref["us"], out["us"].cpu().numpy()[idx])
    xs_bm, J2 = hip.rollout_cost_bm(M, B, hip.pack_bm(out["xs"][:, 0].contiguous()), hip.pack_bm(out["us"]))
    assert eq(J2.cpu().numpy(), J)
    assert eq(hip.unpack_bm(xs_bm, B, (101, 5)).cpu().numpy(), out["xs"].cpu().numpy())
    del xs_bm, J2, small
    # the path the solver API and the benchmark take at this size: launches of COMPACT_CHUNK iterations
    # with the finished problems dropped in between, every output equal to the single launch's
    from ilqr_amd import compaction
    chunked = hip.solve_chunked(M, hip.Opts(), dev(x0), dev(us0), compaction.COMPACT_CHUNK)
    torch.cuda.synchronize()
    for key in ("xs", "us", "J", "k", "K", "mu", "delta", "iters", "status"):
        assert torch.equal(out[key], chunked[key]), key


def test_boxqp1_on_captured_backward_pass_inputs(hip):
    """Every boxQP call (Q_uu, Q_u, bounds, warm start) made by the backward passes of
    real solves, replayed through the in-register m = 1 routine: the early exits that
    replace the reference's long Armijo spin must not change a bit or a counter."""
    import ctypes as C
    L = co.lib()
    cap = 700000
    buf = np.zeros((cap, 5))
    L.oracle_get_qp_capture_count.restype = C.c_long
    L.oracle_set_qp_capture(buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_long(cap))
    try:
        x0, us0 = co.cartpole_batch(48, seed_x=31, seed_u=32)
        co.solve(co.cartpole_model(), co.Opts(), x0, us0, nthreads=1, trace=False)
        n = L.oracle_get_qp_capture_count()
    finally:
        L.oracle_set_qp_capture(None, C.c_long(0))
    qp = buf[:n]
    assert n > 100000
    ref = co.boxqp(qp[:, 0].reshape(-1, 1, 1), qp[:, 1:2], qp[:, 2:3], qp[:, 3:4], qp[:, 4:5], co.Opts())
    assert (ref["n_backtrack"] >= 481).sum() > 100  # the spin really occurs in this workload
    out = hip.boxqp_batched(dev(qp[:, 0].reshape(-1, 1, 1)), dev(qp[:, 1:2]), dev(qp[:, 2:3]), dev(qp[:, 3:4]),
                            dev(qp[:, 4:5]), hip.Opts())
    assert np.array_equal(out["result"].cpu().numpy(), ref["result"])
    assert eq(out["x"].cpu().numpy(), ref["x"])
    assert np.array_equal(out["free"].cpu().numpy().astype(bool), ref["free"])
    cnt = out["counters"].cpu().numpy()
    assert np.array_equal(cnt[:, 0], ref["n_iter"]) and np.array_equal(cnt[:, 1], ref["n_factor"])
    assert np.array_equal(cnt[:, 2], ref["n_backtrack"])
    assert eq(out["Hff"].cpu().numpy()[:, 0, 0], ref["Hff"][:, 0, 0])


def test_workgroup_shapes_agree(hip, monkeypatch):
    """Large batches run 4-wave workgroups, small ones 1-wave workgroups; forcing either on
    the same ragged batch must give the same bits (each wave only talks to itself)."""
    B, T = 700, 40
    Mo = co.cartpole_model(T=T)
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=43, seed_u=44)
    M = hip_model(hip, Mo)
    outs = []
    for blk in ("64", "256"):
        monkeypatch.setenv("ILQR_SOLVE_BLOCK", blk)
        outs.append(hip.solve(M, hip.Opts(max_iter=30), dev(x0), dev(us0)))
    monkeypatch.delenv("ILQR_SOLVE_BLOCK")
    for key in ("xs", "us", "J", "k", "K", "mu", "delta", "iters", "status"):
        assert eq(outs[0][key].cpu().numpy(), outs[1][key].cpu().numpy()), key
    ref = co.solve(Mo, co.Opts(max_iter=30), x0, us0, nthreads=8, trace=False)
    assert eq(outs[1]["us"].cpu().numpy(), ref["us"]) and np.array_equal(outs[1]["iters"].cpu().numpy(), ref["iters"])


def test_dense_and_diagonal_cost_evaluators_agree(hip, monkeypatch):
    """A diagonal Q takes the evaluator that skips the zero products; forcing the
    dense one must give the same bits, and a full Q must match the oracle."""
    B, T = 130, 60
    Mo = co.cartpole_model(T=T)
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=41, seed_u=42)
    M = hip_model(hip, Mo)
    a = hip.solve(M, hip.Opts(max_iter=40), dev(x0), dev(us0))
    monkeypatch.setenv("ILQR_FORCE_DENSE_COST", "1")
    b = hip.solve(M, hip.Opts(max_iter=40), dev(x0), dev(us0))
    monkeypatch.delenv("ILQR_FORCE_DENSE_COST")
    for key in ("xs", "us", "J", "k", "K", "mu", "delta"):
        assert eq(a[key].cpu().numpy(), b[key].cpu().numpy()), key
    assert np.array_equal(a["iters"].cpu().numpy(), b["iters"].cpu().numpy())
    # a full (non-diagonal, non-symmetric) Q and Q_terminal
    r = np.random.RandomState(4)
    Qf = Mo.Q + r.uniform(-2, 2, (5, 5))
    Qtf = Mo.Qt + r.uniform(-2, 2, (5, 5))
    Mf = co.Model(co.MODEL_CARTPOLE, 5, 1, T, co.COST_QR, co.MATH_DET, True, Q=Qf, R=Mo.R, Q_terminal=Qtf,
                  x_goal=Mo.x_goal, u_goal=[0.3], lower=Mo.lower, upper=Mo.upper)
    _check_solve(hip, Mf, x0, us0, 30)


@pytest.mark.parametrize("B,chunk", [(3000, 7), (130, 1), (64, 200)])
def test_chunked_solve_with_compaction_equals_single_launch(hip, B, chunk):
    """Launches of `chunk` iterations with the finished problems dropped in between give the single
    launch's results, counts and per-iteration trace, bit for bit; a sample is checked against the
    oracle as well."""
    from ilqr_amd import compaction
    Mo = co.cartpole_model()
    M = hip_model(hip, Mo)
    x0, us0 = co.cartpole_batch(B, seed_x=71, seed_u=72)
    one = hip.solve(M, hip.Opts(max_iter=60), dev(x0), dev(us0), trace=True)
    chunked = compaction.solve_compacting(hip.solve, M, {}, 60, dev(x0), dev(us0), chunk=chunk, trace=True)
    torch.cuda.synchronize()
    for key in ("xs", "us", "J", "mu", "delta", "iters", "status", "k", "K"):
        assert torch.equal(one[key], chunked[key]), key
    it = one["iters"].cpu().numpy()
    for key in one["trace"]:
        a, b = one["trace"][key].cpu().numpy(), chunked["trace"][key].cpu().numpy()
        for p in range(0, B, max(1, B // 40)):
            assert eq(a[:it[p], p], b[:it[p], p]), (key, p)
    ref = co.solve(Mo, co.Opts(max_iter=60), x0[:40], us0[:40], nthreads=8)
    assert eq(chunked["xs"][:40].cpu().numpy(), ref["xs"]) and np.array_equal(it[:40], ref["iters"])
    assert (it < 60).any() and (it == 60).any()


@pytest.mark.parametrize("B,chunk", [(3000, 7), (100, 1), (70, 500)])
def test_c_level_chunked_solve_equals_single_launch(hip, B, chunk):
    """ilqr_solve_batched_chunked (launches of `chunk` iterations, finished problems dropped in between,
    the loop behind the C ABI) returns what ilqr_solve_batched returns, bit for bit."""
    Mo = co.cartpole_model()
    M = hip_model(hip, Mo)
    x0, us0 = co.cartpole_batch(B, seed_x=73, seed_u=74)
    one = hip.solve(M, hip.Opts(max_iter=60), dev(x0), dev(us0))
    chunked = hip.solve_chunked(M, hip.Opts(max_iter=60), dev(x0), dev(us0), chunk)
    torch.cuda.synchronize()
    for key in ("xs", "us", "J", "mu", "delta", "iters", "status", "k", "K"):
        assert torch.equal(one[key], chunked[key]), key
    it = one["iters"].cpu().numpy()
    assert (it < 60).any() and (it == 60).any()
    # carried solver state: a second call continues from delta/k/K exactly like the single launch
    one2 = hip.solve(M, hip.Opts(max_iter=20), one["xs"][:, 0].contiguous(), one["us"], delta=one["delta"], k=one["k"],
                     K=one["K"])
    ch2 = hip.solve_chunked(M, hip.Opts(max_iter=20), chunked["xs"][:, 0].contiguous(), chunked["us"], 6,
                            delta=chunked["delta"], k=chunked["k"], K=chunked["K"])
    for key in ("xs", "us", "J", "iters", "status", "K"):
        assert torch.equal(one2[key], ch2[key]), key


def test_every_launch_shape_gives_the_same_bits(hip):
    """Problems per wavefront, wavefronts per workgroup, and one, two or three wavefronts per group of problems
    (the backward sweep shared between a master, a helper and a verifier wavefront) only change which lane
    computes what."""
    import os
    Mo = co.cartpole_model(T=40)
    M = hip_model(hip, Mo)
    B = 333  # ragged: the last wavefront and the last workspace tile are partly empty in every shape
    x0, us0 = co.cartpole_batch(B, T=Mo.T, seed_x=91, seed_u=92)
    ref = co.solve(Mo, co.Opts(max_iter=40), x0, us0, nthreads=8)
    names = ("ILQR_SOLVE_LPW", "ILQR_SOLVE_BLOCK", "ILQR_SOLVE_GROUP")
    saved = {k: os.environ.get(k) for k in names}
    try:
        for shape in ((None, None, None), (64, 64, 1), (64, 128, 1), (64, 256, 1), (32, 128, 1), (32, 256, 1),
                      (16, 256, 1), (64, None, 2), (32, None, 2), (16, None, 2), (64, None, 3), (32, None, 3),
                      (16, None, 3)):
            for key, val in zip(names, shape):
                if val is None:
                    os.environ.pop(key, None)
                else:
                    os.environ[key] = str(val)
            out = hip.solve(M, hip.Opts(max_iter=40), dev(x0), dev(us0), trace=True)
            for key in ("xs", "us", "J", "k", "K", "mu", "delta"):
                assert eq(out[key].cpu().numpy(), ref[key]), (shape, key)
            assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"]), shape
            assert np.array_equal(out["status"].cpu().numpy(), ref["status"]), shape
            for key in ("J", "accepted", "alpha_idx", "mu", "delta", "qp_iter", "qp_backtrack", "n_clamped", "stopped_at"):
                tr = out["trace"][key].cpu().numpy().T
                for b in range(0, B, 7):
                    it = int(ref["iters"][b])
                    assert eq(tr[b, :it], ref["trace"][key][b, :it]), (shape, key, b)
    finally:
        for key, val in saved.items():
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val


def test_backward_pass_giving_up_at_mu_max_reuses_the_previous_sums_across_launches(hip):
    """ilqr_core.py:287-313: when the backward pass raises LinAlgError (non-finite Q_uu) and the retry loop
    gives up at mu_max, the line search still runs, on the gains and on the Delta_J1 / Delta_J2 of the
    PREVIOUS iteration.  Forced here with a coarse time step (f_u > 1), a line search that never accepts
    (scale = 1e30) and delta_ = 1e102: mu climbs 1 -> 2e102 -> 8e204 -> 6.4e307, where Q_uu overflows.
    The single launch, launches of one iteration (where the sums have to travel between launches) and the
    C-level chunked solve must all agree with the oracle, including the index at which that last line
    search ended (30 = every alpha tried; 0 would mean the sums were lost)."""
    from ilqr_amd import compaction
    B, T = 70, 6
    Mo = co.cartpole_model(T=T, dt=1.5)
    M = hip_model(hip, Mo)
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=5, seed_u=6)
    x0[:, [0, 1, 4]] *= 0.1
    us0 *= 0.1
    kw = dict(scale=1e30, mu_max=1.5e308)
    d0 = np.full(B, 1e102)
    ref = co.solve(Mo, co.Opts(max_iter=8, **kw), x0, us0, delta=d0, nthreads=8)
    gave_up = (ref["iters"] == 4) & (ref["trace"]["alpha_idx"][:, 3] == 30)
    assert gave_up.sum() > B // 2  # the path really occurs
    one = hip.solve(M, hip.Opts(max_iter=8, **kw), dev(x0), dev(us0), delta=dev(d0), trace=True)
    step = compaction.solve_compacting(hip.solve, M, kw, 8, dev(x0), dev(us0), delta=dev(d0), chunk=1, trace=True)
    for out in (one, step):
        assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
        assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
        for key in ("J", "xs", "us", "k", "K", "mu", "delta"):
            assert eq(out[key].cpu().numpy(), ref[key]), key
        for key in ("alpha_idx", "accepted", "J", "mu", "delta", "stopped_at"):
            tr = out["trace"][key].cpu().numpy().T
            for b in range(B):
                it = int(ref["iters"][b])
                assert eq(tr[b, :it], ref["trace"][key][b, :it]), (key, b)
    for chunk in (1, 3):
        ch = hip.solve_chunked(M, hip.Opts(max_iter=8, **kw), dev(x0), dev(us0), chunk, delta=dev(d0))
        for key in ("xs", "us", "J", "k", "K", "mu", "delta", "iters", "status"):
            assert torch.equal(one[key], ch[key]), (chunk, key)


@pytest.mark.parametrize("group", ["3", "2", "1"])
def test_shared_backward_sweep_on_the_rare_paths(hip, monkeypatch, group):
    """The backward sweep shared by three or two wavefronts (long solves of up to 32768 problems take it by
    default; forced here on short ones) through the paths a swing-up batch rarely visits: the LinAlgError retry loop
    that gives up at mu_max, non-PD breaks (ilqr_core.py:137-140), wrong guesses of the boxQP outcome (the
    warm-started re-solves of a closed loop end their boxQP on other paths than a cold solve), and lanes
    that leave the loop at different iterations."""
    monkeypatch.setenv("ILQR_SOLVE_GROUP", group)  # ("1": the same paths through the one-wavefront sweep)
    # (a) giving up at mu_max, with non-PD breaks on the way (see the test of that name above)
    B, T = 70, 6
    Mo = co.cartpole_model(T=T, dt=1.5)
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=5, seed_u=6)
    x0[:, [0, 1, 4]] *= 0.1
    us0 *= 0.1
    kw = dict(scale=1e30, mu_max=1.5e308)
    d0 = np.full(B, 1e102)
    ref = co.solve(Mo, co.Opts(max_iter=8, **kw), x0, us0, delta=d0, nthreads=8)
    assert (ref["trace"]["stopped_at"] >= 0).any()
    out = hip.solve(hip_model(hip, Mo), hip.Opts(max_iter=8, **kw), dev(x0), dev(us0), delta=dev(d0), trace=True)

    def same(out, ref, B, counters_upto=None):
        assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
        assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
        for key in ("J", "xs", "us", "k", "K", "mu", "delta"):
            assert eq(out[key].cpu().numpy(), ref[key]), key
        for key in ("alpha_idx", "accepted", "J", "mu", "delta", "stopped_at", "qp_iter", "qp_backtrack", "n_clamped"):
            tr = out["trace"][key].cpu().numpy().T
            for b in range(B):
                it = int(ref["iters"][b])
                if counters_upto is not None and key in ("qp_iter", "qp_backtrack", "n_clamped"):
                    it = min(it, counters_upto)
                assert eq(tr[b, :it], ref["trace"][key][b, :it]), (key, b)
    # The boxQP counters of the abandoned sweep (iteration 3) are left out: with mu = 6.4e307 entries of V_xx + mu I
    # overflow to inf in two of the problems, where the reference's dense products turn 0 * inf into NaN one step
    # earlier than a sweep that skips the structural zeros of f_u; the sweep is abandoned either way and nothing
    # of it is used.
    same(out, ref, B, counters_upto=3)
    # (b) the closed-loop configuration (N = 15, force limit 20, pole length 0.5): a cold solve, then a
    # warm-started re-solve from the shifted controls with the carried delta, k, K
    B, T = 300, 15
    Mo = co.cartpole_model(T=T, max_control=20.0, l=0.5, mpc=True)
    M = hip_model(hip, Mo)
    r = np.random.RandomState(77)
    st = np.stack([r.uniform(-0.05, 0.05, B), r.uniform(-0.05, 0.05, B), np.pi / 2 + r.uniform(-0.05, 0.05, B),
                   r.uniform(-0.05, 0.05, B)], axis=1)
    x0, us0 = co.augment_state(st), r.uniform(-1, 1, (B, T, 1))
    ref = co.solve(Mo, co.Opts(max_iter=200), x0, us0, nthreads=8)
    out = hip.solve(M, hip.Opts(max_iter=200), dev(x0), dev(us0), trace=True)
    same(out, ref, B)
    us1 = np.concatenate([ref["us"][:, 1:], ref["us"][:, -1:]], axis=1)
    ref2 = co.solve(Mo, co.Opts(max_iter=10), ref["xs"][:, 1], us1, delta=ref["delta"], k=ref["k"], K=ref["K"], nthreads=8)
    out2 = hip.solve(M, hip.Opts(max_iter=10), dev(ref["xs"][:, 1]), dev(us1), delta=dev(ref["delta"]), k=dev(ref["k"]),
                     K=dev(ref["K"]), trace=True)
    same(out2, ref2, B)
    assert len(np.unique(ref["iters"])) > 3


@pytest.mark.parametrize("group", ["3", "2", "1"])
def test_shortest_horizons_and_option_corners_through_every_sweep_variant(hip, monkeypatch, group):
    """Horizons of one, two and three steps (the shared sweeps hand operands over two steps ahead), a single
    problem, a single step size (no further candidates to try), zero iterations, and a dense cost matrix."""
    monkeypatch.setenv("ILQR_SOLVE_GROUP", group)
    for T, B in ((1, 5), (2, 70), (3, 1), (4, 129)):
        Mo = co.cartpole_model(T=T)
        x0, us0 = co.cartpole_batch(B, T=T, seed_x=40 + T, seed_u=50 + T)
        _check_solve(hip, Mo, x0, us0, 30)
    # one step size only: an iteration whose first rollout is rejected is rejected as a whole
    T, B = 12, 90
    Mo = co.cartpole_model(T=T)
    M = hip_model(hip, Mo)
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=61, seed_u=62)
    for n_alpha in (1, 2):
        kw = dict(max_iter=40, line_search_steps=n_alpha, scale=0.97)  # (a demanding accept ratio: rejections occur)
        ref = co.solve(Mo, co.Opts(**kw), x0, us0, nthreads=8)
        out = hip.solve(M, hip.Opts(**kw), dev(x0), dev(us0), trace=True)
        assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
        for key in ("J", "xs", "us", "k", "K", "mu", "delta"):
            assert eq(out[key].cpu().numpy(), ref[key]), (n_alpha, key)
        assert (ref["trace"]["accepted"] == 0).any()
    # zero iterations: the initial rollout and its cost
    ref = co.solve(Mo, co.Opts(max_iter=0), x0, us0, nthreads=8)
    out = hip.solve(M, hip.Opts(max_iter=0), dev(x0), dev(us0))
    for key in ("J", "xs", "us"):
        assert eq(out[key].cpu().numpy(), ref[key]), key
    assert (out["iters"].cpu().numpy() == 0).all()
    # a full (non-symmetric) Q: the dense cost evaluator in the shared sweeps
    r = np.random.RandomState(7)
    Mf = co.Model(co.MODEL_CARTPOLE, 5, 1, T, co.COST_QR, co.MATH_DET, True, Q=Mo.Q + r.uniform(-2, 2, (5, 5)), R=Mo.R,
                  Q_terminal=Mo.Qt + r.uniform(-2, 2, (5, 5)), x_goal=Mo.x_goal, u_goal=[0.3], lower=Mo.lower,
                  upper=Mo.upper)
    _check_solve(hip, Mf, x0, us0, 25)


@pytest.mark.parametrize("group", ["3", "2", "1"])
def test_a_problem_that_searches_past_the_first_step_size_in_most_iterations(hip, monkeypatch, group):
    """The group of 64 problems of the random 12288-problem batch that holds problem 632: its line search goes past
    alpha_0 in 191 of its 200 iterations (90 winners at a later step size, re-run with stores; 101 iterations in
    which every step size fails) and 98 of its backward passes end on a non-PD Q_uu.  Every output and the whole
    trace against the oracle, through the one-, two- and three-wavefront sweeps."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.setenv("ILQR_SOLVE_GROUP", group)
    x0, us0 = bench.make_batch(12288, 0)
    sl = slice(576, 640)
    ref, _ = _check_solve(hip, co.cartpole_model(T=bench.T), x0[sl], us0[sl], 200)
    it = int(ref["iters"][632 - 576])
    ai = ref["trace"]["alpha_idx"][632 - 576, :it]
    assert it == 200 and (ai > 0).sum() > 150 and (ai == 30).sum() > 50
    assert (ref["trace"]["stopped_at"][632 - 576, :it] >= 0).sum() > 50
