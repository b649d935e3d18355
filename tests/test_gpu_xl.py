"""GPU parity tests of the wide path (ilqr_xl_*: up to 64 states and 16 controls, one problem per workgroup, the
matrices of a backward step resident in LDS) against the CPU oracle's generic solver code on the synthetic model
x+ = x + dt (A x + B u + c phi(W x)).  The reference ships no model of this size (parity unpinned against it for
the model; the machinery is the one pinned by the cart-pole reference runs).  Bar: == on every value and integer."""
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


def pair(hip, n, m, T, u_max=0.6, **kw):
    Mo = co.synthetic_model(n=n, m=m, T=T, seed=40 + n + m, u_max=u_max)
    M = hip.LargeModel(T, Mo.A, Mo.B, Mo.W, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, Mo.u_goal, Mo.lower, Mo.upper, dt=Mo.dt,
                       sig_c=Mo.sig_c, **kw)
    return Mo, M


def compare_solve(out, ref, B, with_trace=True):
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


def compare_backward(bw, ref):
    assert eq(bw["k"].cpu().numpy(), ref["k"]) and eq(bw["K"].cpu().numpy(), ref["K"])
    assert np.array_equal(bw["qp_result"].cpu().numpy(), ref["qp_result"])
    assert np.array_equal(bw["free"].cpu().numpy(), ref["free"])
    info, dJ = bw["info"].cpu().numpy(), bw["dJ"].cpu().numpy()
    assert np.array_equal(info[0], ref["stopped_at"]) and np.array_equal(info[1], ref["qp_iter"])
    assert np.array_equal(info[2], ref["qp_backtrack"]) and np.array_equal(info[3], ref["n_clamped"])
    assert np.array_equal(info[4], ref["rc"])
    assert eq(dJ[0], ref["dJ1"]) and eq(dJ[1], ref["dJ2"])


@pytest.mark.parametrize("n,m", [(64, 16), (48, 12), (33, 9), (40, 16), (64, 3), (17, 2)])
def test_wide_path_bit_exact(hip, n, m):
    """Whole solves with traces, and each stage on its own (rollout + cost, one backward sweep with result codes,
    free sets and counters, a line search with a stored candidate), for sizes on both sides of the tile
    boundaries (the kernels work on 16 x 16 tiles and guard the ragged edge)."""
    T, B, n_it = 40, 24, 15
    Mo, M = pair(hip, n, m, T, wide=True)
    assert M.xl and (M.n, M.m) == (n, m)
    x0, us0 = co.synthetic_batch(B, n=n, m=m, T=T, seed_x=50 + n, seed_u=60 + m)
    us0 *= 2.0
    xs_ref, J_ref = co.rollout_cost(Mo, x0, us0)
    xs, J = hip.large_rollout_cost(M, dev(x0), dev(us0))
    assert eq(xs.cpu().numpy(), xs_ref) and eq(J.cpu().numpy(), J_ref)
    for mu in (1.0, 0.0):
        bw_ref = co.backward(Mo, co.Opts(), xs_ref, us0, np.full(B, mu))
        compare_backward(hip.large_backward(M, hip.Opts(), dev(xs_ref), dev(us0), dev(np.full(B, mu))), bw_ref)
    al = co.Opts().alphas[:9]
    xs_f, us_f, J_f = co.forward(Mo, al, xs_ref, us0, bw_ref["k"], bw_ref["K"])
    Jg, xg, ug = hip.large_forward_ls(M, al, dev(xs_ref), dev(us0), dev(bw_ref["k"]), dev(bw_ref["K"]), store_idx=6)
    assert eq(Jg.cpu().numpy(), J_f) and eq(xg.cpu().numpy(), xs_f[:, 6]) and eq(ug.cpu().numpy(), us_f[:, 6])
    ref = co.solve(Mo, co.Opts(max_iter=n_it), x0, us0, nthreads=8)
    out = hip.large_solve(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    compare_solve(out, ref, B)
    assert (ref["trace"]["n_clamped"] > 0).any()


@pytest.mark.parametrize("T", [1, 2, 4, 7])
def test_wide_path_shortest_horizons(hip, T):
    """Horizons shorter than, and not a multiple of, the depth to which a rollout fetches its operands ahead: whole
    solves and a line search with a stored candidate, for a size with ragged blocks of eight (n = 45, m = 11)."""
    n, m, B = 45, 11, 9
    Mo, M = pair(hip, n, m, T, wide=True)
    x0, us0 = co.synthetic_batch(B, n=n, m=m, T=T, seed_x=70 + T, seed_u=80 + T)
    us0 *= 2.0
    xs_ref, J_ref = co.rollout_cost(Mo, x0, us0)
    xs, J = hip.large_rollout_cost(M, dev(x0), dev(us0))
    assert eq(xs.cpu().numpy(), xs_ref) and eq(J.cpu().numpy(), J_ref)
    bw_ref = co.backward(Mo, co.Opts(), xs_ref, us0, np.full(B, 1.0))
    al = co.Opts().alphas[:6]
    xs_f, us_f, J_f = co.forward(Mo, al, xs_ref, us0, bw_ref["k"], bw_ref["K"])
    Jg, xg, ug = hip.large_forward_ls(M, al, dev(xs_ref), dev(us0), dev(bw_ref["k"]), dev(bw_ref["K"]), store_idx=5)
    assert eq(Jg.cpu().numpy(), J_f) and eq(xg.cpu().numpy(), xs_f[:, 5]) and eq(ug.cpu().numpy(), us_f[:, 5])
    ref = co.solve(Mo, co.Opts(max_iter=10), x0, us0, nthreads=4)
    out = hip.large_solve(M, hip.Opts(max_iter=10), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    compare_solve(out, ref, B)


@pytest.mark.parametrize("n,m", [(32, 8), (20, 5)])
def test_wide_path_equals_the_register_path(hip, n, m):
    """Where both kernel families apply they return the same bits: the model solved on the register-resident
    kernels (zero-padded to 32 x 8) and on the LDS-resident ones (dense)."""
    T, B, n_it = 50, 40, 30
    Mo, M = pair(hip, n, m, T)
    _, Mw = pair(hip, n, m, T, wide=True)
    assert not M.xl and Mw.xl
    x0, us0 = co.synthetic_batch(B, n=n, m=m, T=T, seed_x=3, seed_u=4)
    us0 *= 1.5
    a = hip.large_solve(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0), trace=True)
    b = hip.large_solve(Mw, hip.Opts(max_iter=n_it), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    for key in ("xs", "us", "J", "mu", "delta", "k", "K", "iters", "status"):
        assert torch.equal(a[key], b[key]), key
    for key in a["trace"]:
        assert eq(a["trace"][key].cpu().numpy(), b["trace"][key].cpu().numpy()), key
    assert int(a["iters"].max()) > 5


@pytest.mark.parametrize("qp", [dict(), dict(qp_step_dec=0.6), dict(qp_armijo=0.3, qp_min_step=1e-12),
                                dict(qp_step_dec=0.97, qp_min_step=1e-6)])
def test_wide_backward_on_converged_trajectories(hip, qp):
    """Backward sweeps on trajectories the solver has (nearly) converged on, where the boxQP's Armijo loop runs
    until the step falls below min_step in a large share of the time steps (ilqr_boxQP.py:151-159): the device
    code evaluates the trips 64 at a time, one per lane, over several rounds.  Gains, expected improvements,
    result codes, free sets and trip counters against the oracle's literal loop, for several option settings
    (one of them with more than 400 trips per exhausted search)."""
    n, m, T, B = 48, 12, 80, 24
    Mo, M = pair(hip, n, m, T)
    x0, us0 = co.synthetic_batch(B, n=n, m=m, T=T, seed_x=5, seed_u=6)
    sol = co.solve(Mo, co.Opts(max_iter=12), x0, us0, nthreads=16, trace=False)
    ref = co.backward(Mo, co.Opts(**qp), sol["xs"], sol["us"], sol["mu"])
    compare_backward(hip.large_backward(M, hip.Opts(**qp), dev(sol["xs"]), dev(sol["us"]), dev(sol["mu"])), ref)
    assert (ref["qp_result"] == 2).mean() > 0.02  # the path is really taken


def test_wide_path_rare_paths(hip):
    """A cost whose R is indefinite makes Q_uu non-PD at the end of the horizon until the regularisation has
    grown (the silent break of ilqr_core.py:137-140, rejected iterations, the mu / delta schedule, zero gains for
    the steps not reached); a problem that starts at 1e200 overflows, so every backward pass raises the
    reference's LinAlgError until mu reaches mu_max (:291-299, status 2)."""
    n, m, T, B = 40, 10, 30, 16
    Mq = co.synthetic_model(n=n, m=m, T=T, seed=9, u_max=0.6)
    R = np.array(Mq.R, dtype=float).reshape(m, m).copy()
    R[np.arange(m), np.arange(m)] -= np.linspace(0.0, 2.5, m) * np.diag(R)
    Mo = co.Model(co.MODEL_SIGMOID, n, m, T, co.COST_QR, co.MATH_DET, True, dt=Mq.dt, A=Mq.A, B=Mq.B, W=Mq.W,
                  sig_c=Mq.sig_c, Q=Mq.Q, R=R, Q_terminal=Mq.Qt, x_goal=Mq.x_goal, u_goal=Mq.u_goal, lower=Mq.lower,
                  upper=Mq.upper)
    M = hip.LargeModel(T, Mq.A, Mq.B, Mq.W, Mq.Q, R, Mq.Qt, Mq.x_goal, Mq.u_goal, Mq.lower, Mq.upper, dt=Mq.dt,
                       sig_c=Mq.sig_c)
    x0, us0 = co.synthetic_batch(B, n=n, m=m, T=T, seed_x=7, seed_u=8)
    x0[3] = 1e200
    ref = co.solve(Mo, co.Opts(max_iter=25), x0, us0, nthreads=8)
    out = hip.large_solve(M, hip.Opts(max_iter=25), dev(x0), dev(us0), trace=True)
    torch.cuda.synchronize()
    compare_solve(out, ref, B)
    assert (ref["trace"]["stopped_at"] >= 0).any() and (ref["trace"]["accepted"] == 0).any()
    assert ref["status"][3] == 2
    xs, _ = co.rollout_cost(Mo, x0, us0)
    compare_backward(hip.large_backward(M, hip.Opts(), dev(xs), dev(us0), dev(np.ones(B))),
                     co.backward(Mo, co.Opts(), xs, us0, np.ones(B)))


def test_wide_path_with_one_path_per_problem(hip):
    """PathQRCost with a path per problem (goal strides) and QRCost with a goal per problem on the wide path:
    == a loop of single-problem oracle solves, each with its own goal."""
    n, m, T, B = 40, 10, 25, 12
    Mq = co.synthetic_model(n=n, m=m, T=T, seed=19, u_max=0.6)
    r = np.random.RandomState(23)
    x_path = 0.3 * np.cumsum(r.uniform(-0.1, 0.1, (B, T + 1, n)), axis=1)
    u_path = r.uniform(-0.2, 0.2, (B, T, m))
    x0, us0 = co.synthetic_batch(B, n=n, m=m, T=T, seed_x=31, seed_u=32)
    M = hip.LargeModel(T, Mq.A, Mq.B, Mq.W, Mq.Q, Mq.R, Mq.Qt, None, None, Mq.lower, Mq.upper, dt=Mq.dt,
                       sig_c=Mq.sig_c, x_path=x_path, u_path=u_path)
    out = hip.large_solve(M, hip.Opts(max_iter=12), dev(x0), dev(us0))
    Mg = hip.LargeModel(T, Mq.A, Mq.B, Mq.W, Mq.Q, Mq.R, Mq.Qt, x_path[:, -1], u_path[:, 0], Mq.lower, Mq.upper,
                        dt=Mq.dt, sig_c=Mq.sig_c)
    outg = hip.large_solve(Mg, hip.Opts(max_iter=12), dev(x0), dev(us0))
    torch.cuda.synchronize()
    assert M.xl and Mg.xl
    for b in range(B):
        Mo = co.Model(co.MODEL_SIGMOID, n, m, T, co.COST_PATHQR, co.MATH_DET, True, dt=Mq.dt, A=Mq.A, B=Mq.B, W=Mq.W,
                      sig_c=Mq.sig_c, Q=Mq.Q, R=Mq.R, Q_terminal=Mq.Qt, x_goal=x_path[b], u_goal=u_path[b],
                      lower=Mq.lower, upper=Mq.upper)
        ref = co.solve(Mo, co.Opts(max_iter=12), x0[b:b + 1], us0[b:b + 1], nthreads=1, trace=False)
        for key in ("xs", "us", "J", "K", "iters"):
            assert eq(out[key][b:b + 1].cpu().numpy(), ref[key]), (b, key)
        Mo = co.Model(co.MODEL_SIGMOID, n, m, T, co.COST_QR, co.MATH_DET, True, dt=Mq.dt, A=Mq.A, B=Mq.B, W=Mq.W,
                      sig_c=Mq.sig_c, Q=Mq.Q, R=Mq.R, Q_terminal=Mq.Qt, x_goal=x_path[b, -1], u_goal=u_path[b, 0],
                      lower=Mq.lower, upper=Mq.upper)
        ref = co.solve(Mo, co.Opts(max_iter=12), x0[b:b + 1], us0[b:b + 1], nthreads=1, trace=False)
        for key in ("xs", "us", "J", "K", "iters"):
            assert eq(outg[key][b:b + 1].cpu().numpy(), ref[key]), (b, key)


def test_wide_solve_more_problems_than_workgroups(hip):
    """One workgroup per CU stays resident and takes problems from a queue: with more problems than CUs every
    workgroup solves several, one after the other, on the same workspace slot.  Every output against the oracle,
    and a second call on the same workspace."""
    n, m, T, B = 36, 9, 12, 700
    Mo, M = pair(hip, n, m, T)
    x0, us0 = co.synthetic_batch(B, n=n, m=m, T=T, seed_x=41, seed_u=42)
    ref = co.solve(Mo, co.Opts(max_iter=10), x0, us0, nthreads=16, trace=False)
    out = hip.large_solve(M, hip.Opts(max_iter=10), dev(x0), dev(us0))
    torch.cuda.synchronize()
    compare_solve(out, ref, B, with_trace=False)
    assert ref["iters"].min() < ref["iters"].max()
    again = hip.large_solve(M, hip.Opts(max_iter=10), dev(x0), dev(us0), workspace=out["workspace"])
    assert eq(again["us"].cpu().numpy(), ref["us"])


def test_wide_model_through_the_solver_api(hip):
    """iLQR(SigmoidLinearDynamics(64 states, 16 controls), QRCost, N).update_u on a batch and, with an
    on_iteration callback, one launch per iteration: both equal the oracle's solve."""
    from ilqr_amd import QRCost, iLQR
    from ilqr_amd.synthetic import SigmoidLinearDynamics
    n, m, T, B = 64, 16, 30, 7
    Mo = co.synthetic_model(n=n, m=m, T=T, seed=77, u_max=0.6)
    dyn = SigmoidLinearDynamics(Mo.A, Mo.B, Mo.W, dt=Mo.dt, sig_c=Mo.sig_c, max_control=0.6)
    cost = QRCost(Mo.Q, Mo.R, Q_terminal=Mo.Qt)
    x0, us0 = co.synthetic_batch(B, n=n, m=m, T=T, seed_x=13, seed_u=14)
    us0 *= 2.0
    ref = co.solve(Mo, co.Opts(max_iter=15), x0, us0, nthreads=8)
    s = iLQR(dyn, cost, T)
    xs, us = s.update_u(x0, us0, n_iterations=15)
    assert xs.shape == (B, T + 1, n) and eq(xs, ref["xs"]) and eq(us, ref["us"])
    assert np.array_equal(s.iterations_, ref["iters"]) and eq(s.K_, ref["K"])
    log = []
    t = iLQR(dyn, cost, T)
    _, us1 = t.update_u(x0[0], us0[0], n_iterations=15,
                        on_iteration=lambda it, xs_, us_, J, acc, conv: log.append((it, J, acc, conv)))
    # (the reference leaves the loop on mu >= mu_max before the callback of that iteration, ilqr_core.py:337-345)
    it_n = int(ref["iters"][0]) - (1 if ref["status"][0] == 2 else 0)
    assert eq(us1, ref["us"][0]) and len(log) == it_n
    assert eq([l[1] for l in log], ref["trace"]["J"][0, :it_n])
    assert eq([l[2] for l in log], ref["trace"]["accepted"][0, :it_n] == 1)
