"""GPU parity tests of the large-state path (n = 32, m = 8, one problem per wavefront, fp64 MFMA)
against the CPU oracle.  The reference ships no model of this size (parity unpinned against it);
the oracle's generic solver code is the one pinned by the cart-pole reference runs, applied to
the synthetic model x+ = x + dt (A x + B u + c phi(W x)).  Bar: == on every value (the MFMA
accumulates exactly the oracle's left-to-right fma chains) and on every integer."""
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


def dev(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda:0")


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def models(hip, T):
    Mo = co.synthetic_model(T=T)
    M = hip.LargeModel(T, Mo.A, Mo.B, Mo.W, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, Mo.u_goal, Mo.lower, Mo.upper, dt=Mo.dt,
                       sig_c=Mo.sig_c)
    return Mo, M


def test_large_rollout_cost_bit_exact(hip):
    T, B = 50, 37
    Mo, M = models(hip, T)
    x0, us = co.synthetic_batch(B, T=T)
    us *= 3.0
    xs_ref, J_ref = co.rollout_cost(Mo, x0, us)
    xs, J = hip.large_rollout_cost(M, dev(x0), dev(us))
    assert eq(xs.cpu().numpy(), xs_ref) and eq(J.cpu().numpy(), J_ref)


@pytest.mark.parametrize("scale_u,mu", [(1.0, 1.0), (4.0, 0.0), (2.0, 50.0)])
def test_large_backward_bit_exact(hip, scale_u, mu):
    T, B = 40, 21
    Mo, M = models(hip, T)
    x0, us = co.synthetic_batch(B, T=T, seed_x=7, seed_u=8)
    us *= scale_u
    xs, _ = co.rollout_cost(Mo, x0, us)
    mus = np.full(B, mu)
    ref = co.backward(Mo, co.Opts(), xs, us, mus)
    out = hip.large_backward(M, hip.Opts(), dev(xs), dev(us), dev(mus))
    assert eq(out["k"].cpu().numpy(), ref["k"]) and eq(out["K"].cpu().numpy(), ref["K"])
    dJ, info = out["dJ"].cpu().numpy(), out["info"].cpu().numpy()
    assert eq(dJ[0], ref["dJ1"]) and eq(dJ[1], ref["dJ2"])
    assert np.array_equal(info[0], ref["stopped_at"]) and np.array_equal(info[1], ref["qp_iter"])
    assert np.array_equal(info[2], ref["qp_backtrack"]) and np.array_equal(info[3], ref["n_clamped"])
    assert np.array_equal(out["qp_result"].cpu().numpy(), ref["qp_result"])
    assert np.array_equal(out["free"].cpu().numpy(), ref["free"])
    if scale_u >= 2.0:
        assert (ref["n_clamped"] > 0).any()  # the box limits really bind


def test_large_forward_line_search_bit_exact(hip):
    T, B = 40, 19
    Mo, M = models(hip, T)
    x0, us = co.synthetic_batch(B, T=T, seed_x=9, seed_u=10)
    xs, _ = co.rollout_cost(Mo, x0, us)
    bw = co.backward(Mo, co.Opts(), xs, us, 1.0)
    alphas = co.line_search_alphas(30)
    xs_n, us_n, J_n = co.forward(Mo, alphas, xs, us, bw["k"], bw["K"])
    for store in (0, 5, 21):
        J, xs_new, us_new = hip.large_forward_ls(M, alphas, dev(xs), dev(us), dev(bw["k"]), dev(bw["K"]), store_idx=store)
        assert eq(J.cpu().numpy(), J_n)
        assert eq(xs_new.cpu().numpy(), xs_n[:, store]) and eq(us_new.cpu().numpy(), us_n[:, store])


def test_large_solve_bit_exact(hip):
    T, B = 60, 24
    Mo, M = models(hip, T)
    x0, us0 = co.synthetic_batch(B, T=T, seed_x=11, seed_u=12)
    ref = co.solve(Mo, co.Opts(max_iter=25), x0, us0, nthreads=8)
    out = hip.large_solve(M, hip.Opts(max_iter=25), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    for key in ("J", "mu", "delta", "xs", "us", "k", "K"):
        assert eq(out[key].cpu().numpy(), ref[key]), key
    tr = {key: out["trace"][key].cpu().numpy().T for key in out["trace"]}
    for b in range(B):
        it = int(ref["iters"][b])
        for key in ("J", "accepted", "alpha_idx", "mu", "delta", "qp_iter", "qp_backtrack", "n_clamped", "stopped_at"):
            assert eq(tr[key][b, :it], ref["trace"][key][b, :it]), (b, key)
    assert (ref["trace"]["n_clamped"] > 0).any() and (ref["status"] == 1).any()


@pytest.mark.parametrize("T", [1, 2, 5, 7, 13])
def test_large_solve_shortest_horizons(hip, T):
    """Horizons shorter than, and not a multiple of, the depth to which the first rollout of a line search fetches
    its operands ahead (and of the 16 time steps the sweeps batch): whole solves, every output against the oracle."""
    B = 12
    Mo, M = models(hip, T)
    x0, us0 = co.synthetic_batch(B, T=T, seed_x=61 + T, seed_u=62 + T)
    us0 *= 2.0
    ref = co.solve(Mo, co.Opts(max_iter=12), x0, us0, nthreads=4)
    out = hip.large_solve(M, hip.Opts(max_iter=12), dev(x0), dev(us0))
    torch.cuda.synchronize()
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    for key in ("J", "mu", "delta", "xs", "us", "k", "K"):
        assert eq(out[key].cpu().numpy(), ref[key]), key


def test_large_solve_long_horizon(hip):
    """A horizon longer than the LDS space in which the cost of a stored trajectory keeps its stage costs (1040 steps):
    the costs are then summed round by round.  Whole solves, every output against the oracle."""
    T, B = 1100, 5
    Mo, M = models(hip, T)
    x0, us0 = co.synthetic_batch(B, T=T, seed_x=91, seed_u=92)
    ref = co.solve(Mo, co.Opts(max_iter=5), x0, us0, nthreads=5)
    out = hip.large_solve(M, hip.Opts(max_iter=5), dev(x0), dev(us0))
    torch.cuda.synchronize()
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    for key in ("J", "mu", "delta", "xs", "us", "k", "K"):
        assert eq(out[key].cpu().numpy(), ref[key]), key


def test_large_solve_more_problems_than_wavefronts(hip):
    """The solve kernel keeps one workgroup per CU resident and hands problems out from a queue: with more
    problems than the chip has wavefront slots every wavefront solves several, one after the other, and
    problems with different iteration counts must not disturb each other.  2600 problems (not a multiple
    of four), short horizon, every output against the oracle."""
    T, B = 12, 2600
    Mo, M = models(hip, T)
    x0, us0 = co.synthetic_batch(B, T=T, seed_x=41, seed_u=42)
    ref = co.solve(Mo, co.Opts(max_iter=12), x0, us0, nthreads=16, trace=False)
    out = hip.large_solve(M, hip.Opts(max_iter=12), dev(x0), dev(us0))
    torch.cuda.synchronize()
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    for key in ("J", "mu", "delta", "xs", "us", "k", "K"):
        assert eq(out[key].cpu().numpy(), ref[key]), key
    assert ref["iters"].min() < ref["iters"].max()
    # a second call on the same workspace (the queue's counter is reset by the call)
    again = hip.large_solve(M, hip.Opts(max_iter=12), dev(x0), dev(us0), workspace=out.get("workspace"))
    assert eq(again["us"].cpu().numpy(), ref["us"])


def test_large_model_through_the_solver_api(hip):
    """iLQR(SigmoidLinearDynamics(...), QRCost(...), N).update_u on a batch and, with an
    on_iteration callback, one launch per iteration: both equal the oracle's solve."""
    from ilqr_amd import QRCost, iLQR
    from ilqr_amd.synthetic import SigmoidLinearDynamics
    T, B = 40, 9
    Mo = co.synthetic_model(T=T)
    dyn = SigmoidLinearDynamics(Mo.A, Mo.B, Mo.W, dt=Mo.dt, sig_c=Mo.sig_c, max_control=1.0)
    cost = QRCost(Mo.Q, Mo.R, Q_terminal=Mo.Qt)
    x0, us0 = co.synthetic_batch(B, T=T, seed_x=13, seed_u=14)
    ref = co.solve(Mo, co.Opts(max_iter=20), x0, us0, nthreads=8)
    s = iLQR(dyn, cost, T)
    xs, us = s.update_u(x0, us0, n_iterations=20)
    assert xs.shape == (B, T + 1, 32) and eq(xs, ref["xs"]) and eq(us, ref["us"])
    assert np.array_equal(s.iterations_, ref["iters"]) and eq(s.K_, ref["K"])
    log = []
    t = iLQR(dyn, cost, T)
    xs1, us1 = t.update_u(x0[0], us0[0], n_iterations=20,
                          on_iteration=lambda it, xs_, us_, J, acc, conv: log.append((it, J, acc, conv)))
    it_n = int(ref["iters"][0])
    assert eq(us1, ref["us"][0]) and len(log) == it_n
    assert eq([l[1] for l in log], ref["trace"]["J"][0, :it_n])
    # the device copy of the model is reused while nothing changes, and rebuilt when the cost does
    M1 = s._model()
    assert s._model() is M1
    cost.R = 2.0 * cost.R
    cost._R_plus_R_T = cost.R + cost.R.T
    assert s._model() is not M1
    # host-side f / f_x of the mirror class agree with the oracle's model
    f, fx, fu = co.dynamics(Mo, x0[0], us0[0, 0])
    np.testing.assert_allclose(dyn.f(x0[0], us0[0, 0], 0), f[0], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(dyn.f_x(x0[0], us0[0, 0], 0), fx[0], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(dyn.f_u(x0[0], us0[0, 0], 0), fu[0], rtol=1e-13, atol=1e-13)


def test_large_model_tracks_a_path_with_pathqrcost(hip):
    """PathQRCost (ilqr/ilqr_cost.py:635-770: goals by time step, value without the 1/2) on the
    large-state path, through the solver API, against the oracle's generic code."""
    from ilqr_amd import PathQRCost, iLQR
    from ilqr_amd.synthetic import SigmoidLinearDynamics
    T, B = 30, 10
    Mq = co.synthetic_model(T=T)
    r = np.random.RandomState(21)
    x_path = 0.3 * np.cumsum(r.uniform(-0.1, 0.1, (T + 1, 32)), axis=0)
    u_path = r.uniform(-0.2, 0.2, (T, 8))
    Mo = co.Model(co.MODEL_SIGMOID, 32, 8, T, co.COST_PATHQR, co.MATH_DET, True, dt=Mq.dt, A=Mq.A, B=Mq.B, W=Mq.W,
                  sig_c=Mq.sig_c, Q=Mq.Q, R=Mq.R, Q_terminal=Mq.Qt, x_goal=x_path, u_goal=u_path, lower=Mq.lower,
                  upper=Mq.upper)
    x0, us0 = co.synthetic_batch(B, T=T, seed_x=31, seed_u=32)
    ref = co.solve(Mo, co.Opts(max_iter=15), x0, us0, nthreads=8)
    dyn = SigmoidLinearDynamics(Mq.A, Mq.B, Mq.W, dt=Mq.dt, sig_c=Mq.sig_c, max_control=1.0)
    s = iLQR(dyn, PathQRCost(Mq.Q, Mq.R, x_path, u_path, Q_terminal=Mq.Qt), T)
    xs, us = s.update_u(x0, us0, n_iterations=15)
    assert eq(xs, ref["xs"]) and eq(us, ref["us"]) and eq(s.K_, ref["K"])
    assert np.array_equal(s.iterations_, ref["iters"]) and eq(s.J_, ref["J"])
    # stage level: rollout cost and one backward sweep with the path goals
    Mh = hip.LargeModel(T, Mq.A, Mq.B, Mq.W, Mq.Q, Mq.R, Mq.Qt, None, None, Mq.lower, Mq.upper, dt=Mq.dt,
                        sig_c=Mq.sig_c, x_path=x_path, u_path=u_path)
    xs_ref, J_ref = co.rollout_cost(Mo, x0, us0)
    xs_g, J_g = hip.large_rollout_cost(Mh, dev(x0), dev(us0))
    assert eq(xs_g.cpu().numpy(), xs_ref) and eq(J_g.cpu().numpy(), J_ref)


def _compare_solve(out, ref, B, with_trace=True):
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"])
    for key in ("J", "mu", "delta", "xs", "us", "k", "K"):
        assert eq(out[key].cpu().numpy(), ref[key]), key
    if with_trace:
        tr = {key: out["trace"][key].cpu().numpy().T for key in out["trace"]}
        for b in range(B):
            it = int(ref["iters"][b])
            for key in ("J", "accepted", "alpha_idx", "mu", "delta", "qp_iter", "qp_backtrack", "n_clamped",
                        "stopped_at"):
                assert eq(tr[key][b, :it], ref["trace"][key][b, :it]), (b, key)


def test_large_solve_at_the_configured_horizon(hip):
    """The large-state configuration as stated: T = 200, up to 200 iterations, the default options.  The
    first 16 problems of the benchmark batch, every output and the whole per-iteration trace against the
    oracle."""
    T, B = 200, 16
    Mo, M = models(hip, T)
    x0, us0 = co.synthetic_batch(B, T=T)
    ref = co.solve(Mo, co.Opts(), x0, us0, nthreads=16)
    out = hip.large_solve(M, hip.Opts(), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    _compare_solve(out, ref, B)
    assert (ref["status"] == 1).all() and ref["iters"].max() > 5
    assert (ref["trace"]["n_clamped"] > 0).any()  # the box limits bind on this workload


def test_large_full_size_batch_properties(hip):
    """The benchmark batch of the large-state configuration (8192 problems, T = 200, 200 iterations
    allowed): properties that need no oracle run, a sample of problems against the oracle, and a slice
    solved alone against the same problems inside the batch."""
    from ilqr_amd.synthetic import SyntheticWorkload
    T, B = 200, 8192
    wl = SyntheticWorkload(T=T)
    M = wl.device_model()
    x0, us0 = wl.batch(B)
    out = hip.large_solve(M, hip.Opts(), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    iters, status, J = out["iters"].cpu().numpy(), out["status"].cpu().numpy(), out["J"].cpu().numpy()
    assert np.isfinite(J).all() and iters.min() >= 1 and iters.max() <= 200
    assert set(np.unique(status)) <= {0, 1, 2} and (status == 1).mean() > 0.99
    trJ, acc = out["trace"]["J"].cpu().numpy(), out["trace"]["accepted"].cpu().numpy()
    for b in range(0, B, 61):  # the cost never increases; the last entry is the result
        h = trJ[:iters[b], b]
        assert (np.diff(h) <= 0).all() and h[-1] == J[b]
        assert (acc[iters[b]:, b] == -1).all()
    # the returned states are the rollout of the returned controls, and J is their cost
    xs2, J2 = hip.large_rollout_cost(M, out["xs"][:, 0].contiguous(), out["us"])
    assert torch.equal(xs2, out["xs"]) and eq(J2.cpu().numpy(), J)
    # the slowest problem of the batch and a regular sample, against the oracle
    idx = np.unique(np.concatenate([np.arange(0, B, 373), [int(iters.argmax())]]))
    Mo = co.synthetic_model(T=T)
    ref = co.solve(Mo, co.Opts(), x0[idx], us0[idx], nthreads=16, trace=False)
    assert np.array_equal(ref["iters"], iters[idx]) and np.array_equal(ref["status"], status[idx])
    for key in ("J", "us", "xs", "k", "K", "mu", "delta"):
        assert eq(out[key].cpu().numpy()[idx], ref[key]), key
    # a slice solved alone gives the same bits as inside the big batch (what sharding relies on)
    lo, hi = 4096 + 3, 4096 + 3 + 301
    part = hip.large_solve(M, hip.Opts(), dev(x0[lo:hi]), dev(us0[lo:hi]))
    for key in ("us", "xs", "J", "K"):
        assert torch.equal(part[key], out[key][lo:hi]), key
    assert np.array_equal(part["iters"].cpu().numpy(), iters[lo:hi])


@pytest.mark.parametrize("T,seed", [(25, 17), (60, 18)])
def test_large_backward_pass_is_the_riccati_recursion_and_one_full_step_is_optimal(hip, T, seed):
    """A known answer independent of the oracle and of the reference's code (tests/lq_known_answer.py): with
    c = 0 the synthetic model is linear; with mu = 0 and limits that never bind, one backward pass of the
    device code must give the textbook Riccati gains (1e-9), one alpha = 1 forward pass the optimum of the
    linear-quadratic problem as numpy's dense solve finds it (1e-9 on controls, 1e-10 on the cost), and
    -(Delta_J1 + Delta_J2) the true improvement."""
    import lq_known_answer as lq
    p = lq.problem(T=T, seed=seed)
    big = 1e6
    M = hip.LargeModel(T, p["A"], p["B"], np.zeros((32, 32)), p["Q"], p["R"], p["Qf"], None, None, np.full(8, -big),
                       np.full(8, big), dt=p["dt"], sig_c=0.0, x_path=p["g"], u_path=p["h"])
    x0, us = dev(p["x0"][None]), dev(p["us_start"][None])
    xs, J0 = hip.large_rollout_cost(M, x0, us)
    assert abs(float(J0[0]) - lq.cost(p, lq.rollout(p, p["us_start"]), p["us_start"])) <= 1e-12 * float(J0[0])
    bw = hip.large_backward(M, hip.Opts(), xs, us, dev(np.zeros(1)))
    info = bw["info"].cpu().numpy()
    assert info[0, 0] == -1 and info[3, 0] == 0 and (bw["qp_result"].cpu().numpy() == 5).all()
    J_new, xs_new, us_new = hip.large_forward_ls(M, [1.0], xs, us, bw["k"], bw["K"], store_idx=0)
    dJ = bw["dJ"].cpu().numpy()
    lq.check(p, bw["k"][0].cpu().numpy(), bw["K"][0].cpu().numpy(), dJ[0, 0], dJ[1, 0], us_new[0].cpu().numpy(),
             float(J_new[0, 0]))
    # the whole solve ends on the same optimum
    out = hip.large_solve(M, hip.Opts(max_iter=50, tol=1e-12), x0, us)
    _, _, J_opt = lq.dense_optimum(p)
    assert abs(float(out["J"][0]) - J_opt) <= 1e-9 * J_opt


@pytest.mark.parametrize("path", [False, True])
def test_large_derivative_streams_bit_exact(hip, path):
    """ilqr_large_deriv: f_x, f_u, l_x, l_u, l_xx, l_uu, l_ux per (problem, step) and the terminal l_x, l_xx of
    the synthetic model, against the oracle's derivative evaluation, with QRCost and with PathQRCost; T is
    not a multiple of the kernel's steps-per-workgroup."""
    T, B = 21, 5
    Mq = co.synthetic_model(T=T)
    r = np.random.RandomState(33)
    Qf = Mq.Q + 0.1 * r.randn(32, 32)  # full, non-symmetric cost matrices
    Rf = Mq.R + 0.01 * r.randn(8, 8)
    Qtf = Mq.Qt + 0.1 * r.randn(32, 32)
    x_path, u_path = 0.3 * r.randn(T + 1, 32), 0.2 * r.randn(T, 8)
    if path:
        Mo = co.Model(co.MODEL_SIGMOID, 32, 8, T, co.COST_PATHQR, co.MATH_DET, True, dt=Mq.dt, A=Mq.A, B=Mq.B, W=Mq.W,
                      sig_c=Mq.sig_c, Q=Qf, R=Rf, Q_terminal=Qtf, x_goal=x_path, u_goal=u_path, lower=Mq.lower,
                      upper=Mq.upper)
        M = hip.LargeModel(T, Mq.A, Mq.B, Mq.W, Qf, Rf, Qtf, None, None, Mq.lower, Mq.upper, dt=Mq.dt,
                           sig_c=Mq.sig_c, x_path=x_path, u_path=u_path)
    else:
        xg, ug = 0.1 * r.randn(32), 0.1 * r.randn(8)
        Mo = co.Model(co.MODEL_SIGMOID, 32, 8, T, co.COST_QR, co.MATH_DET, True, dt=Mq.dt, A=Mq.A, B=Mq.B, W=Mq.W,
                      sig_c=Mq.sig_c, Q=Qf, R=Rf, Q_terminal=Qtf, x_goal=xg, u_goal=ug, lower=Mq.lower, upper=Mq.upper)
        M = hip.LargeModel(T, Mq.A, Mq.B, Mq.W, Qf, Rf, Qtf, xg, ug, Mq.lower, Mq.upper, dt=Mq.dt, sig_c=Mq.sig_c)
    x0, us = co.synthetic_batch(B, T=T, seed_x=51, seed_u=52)
    us *= 2.0
    xs, _ = co.rollout_cost(Mo, x0, us)
    out = hip.large_deriv(M, dev(xs), dev(us))
    torch.cuda.synchronize()
    for b in range(B):
        ref = co.derivatives(Mo, xs[b], us[b])
        for key in ("fx", "fu", "lx", "lu", "lxx", "luu", "lux", "lxT", "lxxT"):
            assert eq(out[key][b].cpu().numpy(), ref[key]), (b, key)


@pytest.mark.parametrize("qp", [dict(), dict(qp_step_dec=0.6), dict(qp_armijo=0.3, qp_min_step=1e-12),
                                dict(qp_step_dec=0.97, qp_min_step=1e-6)])
def test_large_backward_on_converged_trajectories(hip, qp):
    """Backward sweeps on trajectories the solver has (nearly) converged on: there the boxQP's Armijo loop runs
    until the step falls below min_step in a large share of the time steps (ilqr_boxQP.py:151-159; the 1e-10
    in its test), the path on which the device code replaces literal evaluation of the trips by a proof of
    their outcome.  Gains, expected improvements, result codes, free sets and the trip counters must equal the
    oracle's literal loop, for several settings of the loop's options."""
    T, B = 200, 40
    Mo, M = models(hip, T)
    x0, us0 = co.synthetic_batch(B, T=T)
    sol = co.solve(Mo, co.Opts(max_iter=9), x0, us0, nthreads=16, trace=False)  # most problems need 8-12
    ref = co.backward(Mo, co.Opts(**qp), sol["xs"], sol["us"], sol["mu"])
    out = hip.large_backward(M, hip.Opts(**qp), dev(sol["xs"]), dev(sol["us"]), dev(sol["mu"]))
    info, dJ = out["info"].cpu().numpy(), out["dJ"].cpu().numpy()
    assert np.array_equal(info[2], ref["qp_backtrack"]) and np.array_equal(info[1], ref["qp_iter"])
    assert np.array_equal(out["qp_result"].cpu().numpy(), ref["qp_result"])
    assert np.array_equal(out["free"].cpu().numpy(), ref["free"])
    assert eq(out["k"].cpu().numpy(), ref["k"]) and eq(out["K"].cpu().numpy(), ref["K"])
    assert eq(dJ[0], ref["dJ1"]) and eq(dJ[1], ref["dJ2"])
    assert np.array_equal(info[0], ref["stopped_at"]) and np.array_equal(info[3], ref["n_clamped"])
    exhausted = (ref["qp_result"] == 2).mean()
    assert exhausted > 0.05, exhausted  # the path is really taken


@pytest.mark.parametrize("n,m", [(16, 4), (24, 8), (32, 3), (9, 1), (20, 5)])
def test_large_path_with_fewer_states_and_controls(hip, n, m):
    """Models between the per-model libraries (n <= 8) and the full 32 x 8: the large-state kernels run them
    zero-padded, and every sum of the reference's arithmetic only gains exact zeros at its end, so the n x m
    oracle solve must come back bit for bit -- trajectories, gains, regularisation state, and the boxQP's result
    codes, free sets and counters (the padded controls must not exist for it: the all-clamped exit, the
    single-free-variable norm and the factorisation count depend on m)."""
    T, B, n_it = 40, 30, 20
    Mo = co.synthetic_model(n=n, m=m, T=T, seed=40 + n + m, u_max=0.6)
    M = hip.LargeModel(T, Mo.A, Mo.B, Mo.W, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, Mo.u_goal, Mo.lower, Mo.upper, dt=Mo.dt,
                       sig_c=Mo.sig_c)
    assert (M.n, M.m) == (n, m)
    x0, us0 = co.synthetic_batch(B, n=n, m=m, T=T, seed_x=50 + n, seed_u=60 + m)
    us0 *= 2.0
    ref = co.solve(Mo, co.Opts(max_iter=n_it), x0, us0, nthreads=8)
    out = hip.large_solve(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    _compare_solve(out, ref, B)
    assert (ref["trace"]["n_clamped"] > 0).any()
    # stage level: one backward sweep with result codes and free sets
    xs, _ = co.rollout_cost(Mo, x0, us0)
    bw_ref = co.backward(Mo, co.Opts(), xs, us0, np.ones(B))
    bw = hip.large_backward(M, hip.Opts(), dev(xs), dev(us0), dev(np.ones(B)))
    assert eq(bw["k"].cpu().numpy(), bw_ref["k"]) and eq(bw["K"].cpu().numpy(), bw_ref["K"])
    assert np.array_equal(bw["qp_result"].cpu().numpy(), bw_ref["qp_result"])
    assert np.array_equal(bw["free"].cpu().numpy(), bw_ref["free"])
    info = bw["info"].cpu().numpy()
    assert np.array_equal(info[1], bw_ref["qp_iter"]) and np.array_equal(info[2], bw_ref["qp_backtrack"])
    assert np.array_equal(info[3], bw_ref["n_clamped"])
    # through the solver API
    from ilqr_amd import QRCost, iLQR
    from ilqr_amd.synthetic import SigmoidLinearDynamics
    dyn = SigmoidLinearDynamics(Mo.A, Mo.B, Mo.W, dt=Mo.dt, sig_c=Mo.sig_c, max_control=0.6)
    s = iLQR(dyn, QRCost(Mo.Q, Mo.R, Q_terminal=Mo.Qt), T)
    xs_a, us_a = s.update_u(x0, us0, n_iterations=n_it)
    assert xs_a.shape == (B, T + 1, n) and eq(xs_a, ref["xs"]) and eq(us_a, ref["us"]) and eq(s.K_, ref["K"])
