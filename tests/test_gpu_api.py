"""GPU tests of the reference-API mirror (iLQR, RecedingHorizonController, boxQP,
CartPoleDynamics, the plant) against the oracle and the recorded reference runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import c_oracle as co  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def example_problem(N=100, mpc=False):
    from ilqr_amd.cartpole import CartPoleDynamics
    from ilqr_amd.ilqr_cost import QRCost
    if mpc:
        dyn = CartPoleDynamics(0.05, max_control=20.0, l=0.5)
    else:
        dyn = CartPoleDynamics(0.05)
    x_goal = dyn.augment_state(np.array([0.0, 0.0, 0.0, 0.0]))
    Q = 100 * np.eye(dyn.state_dimension)
    if mpc:
        Q[0, 0] = Q[1, 1] = 10.0
        Q[4, 4] = 80
        R = np.array([[1.0]])
    else:
        Q[1, 1] = Q[4, 4] = 10.0
        R = np.array([[5.0]])
    cost = QRCost(Q, R, Q_terminal=100 * np.eye(dyn.state_dimension), x_goal=x_goal)
    Mo = co.cartpole_model(T=N, max_control=20.0 if mpc else 5.0, l=0.5 if mpc else 1.0, mpc=mpc)
    return dyn, cost, Mo


def test_example_script_flow_single_problem():
    """examples/cartpole_example.py:11-43 with a seeded us_init, checked against the oracle."""
    from ilqr_amd import iLQR
    dyn, cost, Mo = example_problem()
    N = 100
    x0 = dyn.augment_state(np.array([0.0, 0.0, np.pi, 0.0]))
    us_init = np.random.RandomState(0).uniform(-1, 1, (N, dyn.control_dimension))
    ilqr = iLQR(dyn, cost, N)
    xs, us = ilqr.update_u(x0, us_init)
    ref = co.solve(Mo, co.Opts(), x0, us_init)
    assert xs.shape == (N + 1, 5) and us.shape == (N, 1) and isinstance(xs, np.ndarray)
    assert eq(xs, ref["xs"][0]) and eq(us, ref["us"][0])
    assert eq(ilqr.k_, ref["k"][0]) and eq(ilqr.K_, ref["K"][0])
    assert ilqr.mu_ == ref["mu"][0] and ilqr.delta_ == ref["delta"][0]
    assert ilqr.iterations_ == ref["iters"][0] and ilqr.J_ == ref["J"][0]
    assert eq(ilqr.nominal_xs_, xs) and eq(ilqr.nominal_us_, us)
    # a second call on the same object continues from delta_, k_, K_ (mu_ is reset)
    xs2, us2 = ilqr.update_u(xs[1], np.vstack([us[1:], us[-1:]]), n_interations=10)
    ref2 = co.solve(Mo, co.Opts(max_iter=10), xs[1], np.vstack([us[1:], us[-1:]]), delta=ref["delta"], k=ref["k"],
                    K=ref["K"])
    assert eq(xs2, ref2["xs"][0]) and eq(us2, ref2["us"][0])


def test_on_iteration_callback_matches_single_launch_and_oracle():
    from ilqr_amd import iLQR
    dyn, cost, Mo = example_problem(N=60)
    x0, us0 = co.cartpole_batch(1, T=60, seed_x=51, seed_u=52)
    ref = co.solve(Mo, co.Opts(max_iter=40), x0, us0)
    it_n = int(ref["iters"][0])
    log = []

    def on_iteration(iteration, xs, us, J_opt, accepted, converged):
        log.append((iteration, xs.copy(), us.copy(), J_opt, accepted, converged))
    a = iLQR(dyn, cost, 60)
    xs_a, us_a = a.update_u(x0[0], us0[0], n_interations=40, on_iteration=on_iteration)
    b = iLQR(dyn, cost, 60)
    xs_b, us_b = b.update_u(x0[0], us0[0], n_interations=40)
    assert eq(xs_a, xs_b) and eq(us_a, us_b) and eq(xs_a, ref["xs"][0])
    assert eq(a.k_, b.k_) and eq(a.K_, b.K_) and a.mu_ == b.mu_ and a.delta_ == b.delta_
    assert a.iterations_ == b.iterations_ == it_n
    assert [l[0] for l in log] == list(range(it_n))
    assert eq([l[3] for l in log], ref["trace"]["J"][0, :it_n])
    assert [l[4] for l in log] == [bool(v) for v in ref["trace"]["accepted"][0, :it_n]]
    assert [l[5] for l in log][-1] == bool(ref["converged"][0]) and not any(l[5] for l in log[:-1])
    assert eq(log[-1][1], xs_a) and log[0][1].shape == (61, 5)


def test_batched_update_u_numpy_and_torch():
    from ilqr_amd import iLQR
    dyn, cost, Mo = example_problem(N=50)
    B = 100
    x0, us0 = co.cartpole_batch(B, T=50, seed_x=61, seed_u=62)
    ref = co.solve(Mo, co.Opts(max_iter=30), x0, us0, nthreads=8, trace=False)
    s = iLQR(dyn, cost, 50)
    xs, us = s.update_u(x0, us0, n_iterations=30)
    assert xs.shape == (B, 51, 5) and eq(xs, ref["xs"]) and eq(us, ref["us"])
    assert eq(s.delta_, ref["delta"]) and eq(s.mu_, ref["mu"]) and eq(s.K_, ref["K"])
    assert np.array_equal(s.iterations_, ref["iters"]) and np.array_equal(s.status_, ref["status"])
    t = iLQR(dyn, cost, 50)
    xs_t, us_t = t.update_u(torch.tensor(x0, device="cuda"), torch.tensor(us0, device="cuda"), n_iterations=30)
    assert xs_t.is_cuda and eq(xs_t.cpu().numpy(), ref["xs"]) and eq(us_t.cpu().numpy(), ref["us"])
    with pytest.raises(ValueError):
        s.update_u(x0[:7], us0[:7])  # the solver state holds 100 problems
    with pytest.raises(AssertionError):
        iLQR(dyn, cost, 50).update_u(x0, us0[:, :49])


def test_update_u_on_a_batch_that_fills_the_gpu_is_the_single_launch():
    """From 49152 problems on update_u solves in chunks of iterations and drops the finished problems
    in between: same results as the C entry point called once, and a sample equals the oracle."""
    from ilqr_amd import _hip, compaction, iLQR
    dyn, cost, Mo = example_problem(N=30)
    B = compaction.COMPACT_MIN_BATCH
    x0, us0 = co.cartpole_batch(B, T=30, seed_x=81, seed_u=82)
    dx, du = torch.tensor(x0, device="cuda"), torch.tensor(us0, device="cuda")
    s = iLQR(dyn, cost, 30)
    xs, us = s.update_u(dx, du, n_iterations=80)
    M = _hip.Model(30, cost.Q, cost.R, cost.Q_terminal, cost.x_goal, None, [-5.0], [5.0])
    one = _hip.solve(M, _hip.Opts(max_iter=80), dx, du)
    assert torch.equal(xs, one["xs"]) and torch.equal(us, one["us"]) and torch.equal(s.K_, one["K"])
    assert torch.equal(s.iterations_, one["iters"]) and torch.equal(s.status_, one["status"])
    it = one["iters"].cpu().numpy()
    assert it.min() < compaction.COMPACT_CHUNK < it.max()
    pick = np.linspace(0, B - 1, 24).astype(int)
    ref = co.solve(Mo, co.Opts(max_iter=80), x0[pick], us0[pick], nthreads=8, trace=False)
    assert eq(xs[pick].cpu().numpy(), ref["xs"]) and np.array_equal(it[pick], ref["iters"])


def test_receding_horizon_closed_loop_against_reference_run(golden_dir):
    """RecedingHorizonController + plant as in examples/cartpole_example_mpc.py:66-79, against
    the run recorded from the reference's controller (which includes re-solves that hit a
    non-PD Q_uu), and bit for bit against the oracle on a batch."""
    from ilqr_amd import RecedingHorizonController, iLQR
    from ilqr_amd.cartpole_gym import CartPoleContinuousEnv
    g = np.load(golden_dir + "/ref_mpc.npz")
    N, steps = int(g["N"]), int(g["steps"])
    dyn, cost, Mo = example_problem(N=N, mpc=True)
    env = CartPoleContinuousEnv()
    assert (env.length, env.max_force, env.tau) == (0.5, 20.0, 0.05)
    obs = env.reset()
    assert eq(obs, g["st0"])
    ctrl = RecedingHorizonController(dyn.augment_state(obs), g["us_init"].copy(), iLQR(env.dynamics, cost, N), env)
    gen = ctrl.control()
    st = obs
    for s in range(steps):
        xs, us = next(gen)
        assert xs.shape == (2, 5) and us.shape == (1, 1)
        np.testing.assert_allclose(us[0, 0], g["u_applied"][s], rtol=1e-7, atol=1e-9)
        st, _, _, _ = env.step(us[0])
        np.testing.assert_allclose(st, g["states"][s + 1], rtol=1e-7, atol=1e-9)
        assert ctrl.controller_.mu_ == g["mu"][s] and ctrl.controller_.delta_ == g["delta"][s]
        ctrl.set_state(dyn.augment_state(st))

    # batch of closed loops, bit-exact against the oracle (same deterministic math)
    B, steps = 48, 12
    r = np.random.RandomState(3)
    st = np.tile(np.array([0.0, 0.0, 0.5 * np.pi, 0.0]), (B, 1)) + r.uniform(-0.05, 0.05, (B, 4))
    us_init = r.uniform(-1, 1, (B, N, 1))
    env.state = st.copy()
    ctrl = RecedingHorizonController(dyn.augment_state(st), us_init.copy(), iLQR(env.dynamics, cost, N), env)
    gen = ctrl.control(step_size=1, initial_n_iterations=50, subsequent_n_iterations=10)
    o_st, o_us, o_delta, o_k, o_K, n_it = st.copy(), us_init.copy(), None, None, None, 50
    for s in range(steps):
        xs, us = next(gen)
        ref = co.solve(Mo, co.Opts(max_iter=n_it), co.augment_state(o_st), o_us, delta=o_delta, k=o_k, K=o_K,
                       nthreads=8, trace=False)
        assert eq(us[:, 0], ref["us"][:, 0]) and eq(xs[:, 1], ref["xs"][:, 1])
        o_delta, o_k, o_K = ref["delta"], ref["k"], ref["K"]
        o_st = co.plant_step(o_st, ref["us"][:, 0, 0])
        o_us = np.concatenate([ref["us"][:, 1:], ref["us"][:, -1:]], axis=1)
        n_it = 10
        st, _, done, _ = env.step(us[:, 0])
        assert eq(st, o_st) and done.shape == (B,)
        ctrl.set_state(dyn.augment_state(st))


def test_mpc_full_size_closed_loops_on_device():
    """4096 closed loops (the MPC configuration of the examples: N=15, 200 then 10 iterations
    per control step, plant in the loop), tensors resident on the GPU.  Sampled loops must equal
    the oracle's closed loop bit for bit (same deterministic sin/cos in augment_state), a slice
    of the batch must equal the batch, and the loops must stay bounded."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import mpc_closed_loop as mpc
    B, steps, N = 4096, 40, 15
    res = mpc.run(B=B, steps=steps, N=N, record=True)
    assert res["states"].shape == (steps + 1, B, 4) and np.isfinite(res["states"]).all()
    assert res["iterations"][0].max() <= 200 and res["iterations"][1:].max() <= 10
    # (with this horizon and force limit the example's controller does not swing the pole up: like
    # the recorded reference run, the loops settle near theta = pi with the cart within a metre or two)
    assert np.abs(res["final_state"][:, 0]).max() < 5.0 and np.abs(res["final_state"][:, 2] - np.pi).max() < 1.0

    def oracle_augment(st):
        s, c, _ = co.detmath(st[:, 2], st[:, 2])
        return np.stack([st[:, 0], st[:, 1], s, c, st[:, 3]], axis=1)
    pick = np.r_[0:8, B - 8:B]
    st_all, us_all = mpc.initial_conditions(B, N)
    st, us = st_all[pick], us_all[pick]
    Mo = co.cartpole_model(T=N, max_control=20.0, l=0.5, mpc=True)
    delta = k = K = None
    n_it = 200
    for s_ in range(steps):
        ref = co.solve(Mo, co.Opts(max_iter=n_it), oracle_augment(st), us, delta=delta, k=k, K=K, nthreads=8,
                       trace=False)
        assert eq(res["controls"][s_][pick], ref["us"][:, 0, 0]), s_
        assert np.array_equal(res["iterations"][s_][pick], ref["iters"])
        delta, k, K = ref["delta"], ref["k"], ref["K"]
        st = co.plant_step(st, ref["us"][:, 0, 0])
        assert eq(res["states"][s_ + 1][pick], st)
        us = np.concatenate([ref["us"][:, 1:], ref["us"][:, -1:]], axis=1)
        n_it = 10


def test_graphed_closed_loop_equals_the_eager_controller():
    """The control step captured once in a HIP graph and replayed: states and applied controls of
    every step equal those of RecedingHorizonController + set_state, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import mpc_closed_loop as mpc
    B, steps = 300, 25
    eager = mpc.run(B=B, steps=steps, record=True)
    graphed = mpc.run_graphed(B=B, steps=steps, record=True)
    assert eq(graphed["states"], eager["states"]) and eq(graphed["controls"], eager["controls"])


def test_mpc_advance_equals_plant_step_augment_and_shift():
    """The one-launch step between two solves of a closed loop against the oracle's plant step, the
    oracle's augmentation and a numpy shift, bit for bit; aliasing the seed with the solved sequence is refused."""
    import ctypes
    from ilqr_amd import _hip
    DEV = "cuda:0"

    def dev(a):
        return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)

    def oracle_augment(st_):
        s_, c_, _ = co.detmath(st_[:, 2], st_[:, 2])
        return np.stack([st_[:, 0], st_[:, 1], s_, c_, st_[:, 3]], axis=1)
    r = np.random.RandomState(17)
    B, N = 777, 15
    st = np.concatenate([r.uniform(-2, 2, (B, 2)), r.uniform(-4, 4, (B, 1)), r.uniform(-3, 3, (B, 1))], axis=1)
    us = r.uniform(-20, 20, (B, N, 1))
    d_st, d_us = dev(st), dev(us)
    d_seed, d_u, d_x0 = torch.empty_like(d_us), torch.empty(B, dtype=torch.float64, device=DEV), \
        torch.empty((B, 5), dtype=torch.float64, device=DEV)
    _hip.mpc_advance(d_st, d_us, d_seed, d_u, d_x0)
    st_ref = co.plant_step(st, us[:, 0, 0])
    assert eq(d_u.cpu().numpy(), us[:, 0, 0]) and eq(d_st.cpu().numpy(), st_ref)
    assert eq(d_x0.cpu().numpy(), oracle_augment(st_ref))
    assert eq(d_seed.cpu().numpy(), np.concatenate([us[:, 1:], us[:, -1:]], axis=1))
    par = (ctypes.c_double * 5)(9.8, 1.0, 0.1, 0.5, 0.05)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert _hip.lib().ilqr_cartpole_mpc_advance(ctypes.c_int64(B), ctypes.c_int64(N), par, p(d_st), p(d_us), p(d_us), p(d_u),
                                                p(d_x0), None) == -1


def test_pathqrcost_through_the_solver_api(golden_dir):
    """iLQR(dynamics, PathQRCost(...)) as a user of the reference would write it; single problem
    against the recorded reference run, batch against the oracle."""
    from ilqr_amd import PathQRCost, iLQR
    from ilqr_amd.cartpole import CartPoleDynamics
    g = np.load(golden_dir + "/ref_pathqr_solve.npz")
    N = int(g["N"])
    dyn = CartPoleDynamics(0.05)
    cost = PathQRCost(g["Q"], g["R"], g["x_path"], g["u_path"], Q_terminal=g["Qt"])
    s = iLQR(dyn, cost, N)
    xs, us = s.update_u(g["p2_x0"], g["p2_us_init"], n_interations=int(g["p2_max_iter"]))
    assert s.iterations_ == len(g["p2_it_J"]) and s.status_ == 1
    np.testing.assert_allclose(s.J_, g["p2_it_J"][-1], rtol=1e-9)
    np.testing.assert_allclose(us, g["p2_us"], rtol=0, atol=1e-8)
    Mo = co.Model(co.MODEL_CARTPOLE, 5, 1, N, co.COST_PATHQR, co.MATH_DET, True, dt=0.05, Q=g["Q"], R=g["R"],
                  Q_terminal=g["Qt"], x_goal=g["x_path"], u_goal=g["u_path"], lower=[-5.0], upper=[5.0])
    B = 70
    x0, us0 = co.cartpole_batch(B, T=N, seed_x=91, seed_u=92)
    ref = co.solve(Mo, co.Opts(max_iter=25), x0, us0, nthreads=8, trace=False)
    xs_b, us_b = iLQR(dyn, cost, N).update_u(x0, us0, n_iterations=25)
    assert eq(xs_b, ref["xs"]) and eq(us_b, ref["us"])
    with pytest.raises(ValueError):
        iLQR(dyn, cost, N + 1)  # the path is too short for this horizon
    # A caller that moves the path between two calls (receding-horizon tracking shifts the goals,
    # ilqr_core.py:437-438) must get the new path on the device: same solver object, path shifted one step.
    s2 = iLQR(dyn, cost, N)
    s2.update_u(x0, us0, n_iterations=5)
    cost.x_path = np.concatenate([cost.x_path[1:], cost.x_path[-1:]])
    cost.u_path = np.concatenate([cost.u_path[1:], cost.u_path[-1:]])
    Mo2 = co.Model(co.MODEL_CARTPOLE, 5, 1, N, co.COST_PATHQR, co.MATH_DET, True, dt=0.05, Q=g["Q"], R=g["R"],
                   Q_terminal=g["Qt"], x_goal=cost.x_path, u_goal=cost.u_path, lower=[-5.0], upper=[5.0])
    ref2 = co.solve(Mo2, co.Opts(max_iter=12), x0, us0, nthreads=8, trace=False)
    xs_2, us_2 = iLQR(dyn, cost, N).update_u(x0, us0, n_iterations=12)   # fresh solver state, same cost object
    assert eq(xs_2, ref2["xs"]) and eq(us_2, ref2["us"])
    assert not eq(us_2, ref["us"])


def test_boxqp_api(golden_dir):
    from ilqr_amd.ilqr_boxQP import boxQP
    g = np.load(golden_dir + "/ref_boxqp.npz")
    for i in range(int(g["n_cases"])):
        p = "q%d_" % i
        x0 = g[p + "x0"] if bool(g[p + "has_x0"]) else None
        kw = {'maxIter': 100, 'minGrad': 1e-8, 'minRelImprove': 1e-8, 'stepDec': float(g[p + "step_dec"]),
              'minStep': 1e-22, 'Armijo': 0.1, 'verbose': 0}
        x, result, Hff, free = boxQP(g[p + "H"], g[p + "g"], g[p + "lower"], g[p + "upper"], x0, **kw)
        assert result == int(g[p + "result"]) and np.array_equal(free, g[p + "free"]), i
        np.testing.assert_allclose(x, g[p + "x"], rtol=1e-9, atol=1e-12)
        assert Hff.shape == g[p + "Hff"].shape, i
        np.testing.assert_allclose(np.tril(Hff), np.tril(g[p + "Hff"]), rtol=1e-9, atol=1e-12)
    # default options are the reference's (stepDec 0.6), and the batch form
    r = np.random.RandomState(0)
    A = r.randn(40, 3, 3)
    H = np.einsum("bij,bkj->bik", A, A) + 0.1 * np.eye(3)
    g_, lo, up = r.randn(40, 3), -np.ones((40, 3)), np.ones((40, 3))
    x, result, Hff, free = boxQP(H, g_, lo, up)
    ref = co.boxqp(H, g_, lo, up, None, co.Opts(qp_step_dec=0.6))
    assert eq(x, ref["x"]) and np.array_equal(result, ref["result"]) and np.array_equal(free, ref["free"])
    with pytest.raises(np.linalg.LinAlgError):
        boxQP(np.array([[np.nan, 0.0], [0.0, 1.0]]), np.ones(2), -np.ones(2), np.ones(2), np.zeros(2))
    with pytest.raises(AssertionError):
        boxQP(np.eye(2), np.ones(2), -np.ones(2), np.array([1.0, np.inf]))


def test_cartpole_dynamics_point_evaluations():
    from ilqr_amd.cartpole import CartPoleDynamics
    dyn = CartPoleDynamics(0.05, max_control=7.0, mc=1.3, mp=0.2, l=0.7, g=9.81)
    Mo = co.Model(co.MODEL_CARTPOLE, 5, 1, 1, co.COST_QR, co.MATH_DET, True, dt=0.05, mc=1.3, mp=0.2, l=0.7, g=9.81,
                  Q=np.eye(5), R=np.eye(1), lower=[-7.0], upper=[7.0])
    r = np.random.RandomState(2)
    x, u = r.uniform(-2, 2, (30, 5)), r.uniform(-7, 7, (30, 1))
    f, fx, fu = co.dynamics(Mo, x, u)
    assert eq(dyn.f(x, u, 0), f) and eq(dyn.f_x(x, u, 0), fx) and eq(dyn.f_u(x, u, 0), fu)
    assert eq(dyn.f(x[3], u[3], 0), f[3]) and dyn.f_x(x[3], u[3], 0).shape == (5, 5)
    assert dyn.f_u(x[3], u[3], 0).shape == (5, 1)
    assert np.array_equal(dyn.upper_bound, [7.0]) and np.array_equal(dyn.lower_bound, [-7.0])


def test_a_fresh_solve_leaves_the_carry_array_alone():
    """ilqr_opts.dJ_carry belongs to solves that are being continued (mu_init >= 0): a fresh solve
    (mu_init < 0) given a non-NULL pointer neither reads nor writes it -- on the cart-pole path and on the
    large-state path -- and a continued one does write it."""
    from ilqr_amd import _hip
    from ilqr_amd.synthetic import SyntheticWorkload
    dev = torch.device("cuda:0")
    B, T = 70, 30
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=5, seed_u=6)
    Mo = co.cartpole_model(T=T)
    M = _hip.Model(T, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, Mo.u_goal, Mo.lower, Mo.upper)
    x0d, us0d = torch.tensor(x0, device=dev), torch.tensor(us0, device=dev)
    canary = torch.full((2, B), 12345.678, dtype=torch.float64, device=dev)
    dJ = canary.clone()
    plain = _hip.solve(M, _hip.Opts(max_iter=6), x0d, us0d)
    fresh = _hip.solve(M, _hip.Opts(max_iter=6), x0d, us0d, dJ=dJ)
    assert torch.equal(dJ, canary)
    assert torch.equal(fresh["us"], plain["us"]) and torch.equal(fresh["J"], plain["J"])
    cont = _hip.solve(M, _hip.Opts(max_iter=2, mu_from_array=True), x0d, fresh["us"], delta=fresh["delta"], k=fresh["k"],
                      K=fresh["K"], mu=fresh["mu"].clone(), dJ=dJ)
    assert not torch.equal(dJ, canary) and int(cont["iters"].max()) >= 1

    wl = SyntheticWorkload(T=12)
    ML = wl.device_model()
    lx0, lus = wl.batch(9)
    lx0, lus = torch.tensor(lx0, device=dev), torch.tensor(lus, device=dev)
    dJ = torch.full((2, 9), 12345.678, dtype=torch.float64, device=dev)
    _hip.large_solve(ML, _hip.Opts(max_iter=3), lx0, lus, dJ=dJ)
    assert torch.equal(dJ, torch.full_like(dJ, 12345.678))
