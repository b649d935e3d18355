"""GPU tests of goals that differ from problem to problem: every problem of a batch with its own QRCost goal
or its own PathQRCost path.  The reference solves one problem per solver object, each with its own cost
(ilqr/ilqr_cost.py:490-530, :639-770); the batch must therefore equal a Python loop of single-problem solves
of the oracle, each with that problem's goal, bit for bit."""
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
KEYS = ("xs", "us", "J", "k", "K", "mu", "delta")


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def cartpole_goals(B, seed):
    r = np.random.RandomState(seed)
    st = np.zeros((B, 4))
    st[:, 0] = r.uniform(-1.5, 1.5, B)  # where the cart should stand, pole up
    return co.augment_state(st), r.uniform(-0.3, 0.3, (B, 1))


def cartpole_paths(B, T, seed):
    r = np.random.RandomState(seed)
    st = np.zeros((B, T + 1, 4))
    st[:, :, 0] = np.cumsum(r.uniform(-0.03, 0.03, (B, T + 1)), axis=1) + r.uniform(-1, 1, (B, 1))
    st[:, :, 2] = r.uniform(-0.2, 0.2, (B, 1)) * np.linspace(1.0, 0.0, T + 1)[None]
    return co.augment_state(st), r.uniform(-0.2, 0.2, (B, T, 1))


def oracle_loop(make_model, x0, us0, max_iter):
    """One oracle solve per problem, each with its own model object (= its own goal)."""
    outs = [co.solve(make_model(b), co.Opts(max_iter=max_iter), x0[b:b + 1], us0[b:b + 1], trace=False)
            for b in range(x0.shape[0])]
    return {key: np.concatenate([o[key] for o in outs]) for key in KEYS + ("iters", "status")}


def compare(out, ref, what=""):
    assert np.array_equal(out["iters"].cpu().numpy(), ref["iters"]), what
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"]), what
    for key in KEYS:
        assert eq(out[key].cpu().numpy(), ref[key]), (what, key)


def hip_cartpole(hip, Mo, **goals):
    p = Mo.params
    return hip.Model(Mo.T, Mo.Q, Mo.R, Mo.Qt, lower=Mo.lower, upper=Mo.upper, dt=p["dt"], mc=p["mc"], mp=p["mp"],
                     l=p["l"], g=p["g"], **goals)


def cart_model(T, path=None, **kw):
    base = co.cartpole_model(T=T)
    if path is None:
        return co.Model(co.MODEL_CARTPOLE, 5, 1, T, co.COST_QR, co.MATH_DET, True, Q=base.Q, R=base.R, Q_terminal=base.Qt,
                        lower=base.lower, upper=base.upper, **kw)
    return co.Model(co.MODEL_CARTPOLE, 5, 1, T, co.COST_PATHQR, co.MATH_DET, True, Q=base.Q, R=base.R,
                    Q_terminal=base.Qt, lower=base.lower, upper=base.upper, x_goal=path[0], u_goal=path[1])


def test_cartpole_every_problem_its_own_goal(hip):
    T, B, n_it = 40, 150, 30
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=31, seed_u=32)
    xg, ug = cartpole_goals(B, 33)
    ref = oracle_loop(lambda b: cart_model(T, x_goal=xg[b], u_goal=ug[b]), x0, us0, n_it)
    assert len(set(ref["iters"].tolist())) > 3  # the goals really change the solves
    M = hip_cartpole(hip, co.cartpole_model(T=T), x_goal=xg, u_goal=ug)
    compare(hip.solve(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0)), ref, "single launch")
    compare(hip.solve_chunked(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0), 4), ref, "chunked, finished problems dropped")
    # the entry point that allocates nothing
    f64 = torch.float64
    out = dict(xs=torch.empty((B, T + 1, 5), dtype=f64, device=DEV), us=torch.empty((B, T, 1), dtype=f64, device=DEV),
               J=torch.empty(B, dtype=f64, device=DEV), mu=torch.empty(B, dtype=f64, device=DEV),
               iters=torch.empty(B, dtype=torch.int32, device=DEV), status=torch.empty(B, dtype=torch.int32, device=DEV))
    delta = torch.full((B,), 2.0, dtype=f64, device=DEV)
    k, K = torch.zeros((B, T, 1), dtype=f64, device=DEV), torch.zeros((B, T, 1, 5), dtype=f64, device=DEV)
    hip.solve_static(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0), delta, k, K, out)
    out.update(delta=delta, k=k, K=K)
    compare(out, ref, "static workspace")
    # a sub-batch takes its goals along
    idx = torch.tensor([5, 17, 18, 140], device=DEV)
    part = hip.solve(M.subset(idx), hip.Opts(max_iter=n_it), dev(x0)[idx], dev(us0)[idx])
    assert eq(part["us"].cpu().numpy(), ref["us"][idx.cpu().numpy()])
    with pytest.raises(ValueError):
        hip.solve(M, hip.Opts(max_iter=2), dev(x0[:7]), dev(us0[:7]))
    # a shared goal given once per problem is the shared-goal solve
    same = hip_cartpole(hip, co.cartpole_model(T=T), x_goal=np.tile(xg[:1], (B, 1)), u_goal=np.tile(ug[:1], (B, 1)))
    Ms = hip.Model(T, M.Q, M.R, M.Qt, xg[0], ug[0], M.lower, M.upper)
    a, b = hip.solve(same, hip.Opts(max_iter=n_it), dev(x0), dev(us0)), hip.solve(Ms, hip.Opts(max_iter=n_it), dev(x0), dev(us0))
    for key in KEYS + ("iters", "status"):
        assert eq(a[key].cpu().numpy(), b[key].cpu().numpy()), key


@pytest.mark.parametrize("group", ["1", "2", "3"])
def test_cartpole_every_problem_its_own_path(hip, group, monkeypatch):
    """... through every sweep variant (one, two or three wavefronts per group of problems), with enough
    iterations for line searches that go past the first step size (those evaluate OTHER lanes' problems)."""
    monkeypatch.setenv("ILQR_SOLVE_GROUP", group)
    T, B, n_it = 30, 130, 40
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=41, seed_u=42)
    xp, up = cartpole_paths(B, T, 43)
    ref = oracle_loop(lambda b: cart_model(T, path=(xp[b], up[b])), x0, us0, n_it)
    M = hip_cartpole(hip, co.cartpole_model(T=T), x_path=xp, u_path=up)
    out = hip.solve(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0), trace=True)
    compare(out, ref, "group " + group)
    assert int((out["trace"]["alpha_idx"] > 0).sum()) > 0  # searches past alpha_0 occurred


def test_goals_through_the_solver_api_and_the_callback_mode(hip):
    from ilqr_amd import PathQRCost, QRCost, iLQR
    from ilqr_amd.cartpole import CartPoleDynamics
    T, B, n_it = 25, 40, 20
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=51, seed_u=52)
    base = co.cartpole_model(T=T)
    xg, ug = cartpole_goals(B, 53)
    ref = oracle_loop(lambda b: cart_model(T, x_goal=xg[b], u_goal=ug[b]), x0, us0, n_it)
    s = iLQR(CartPoleDynamics(0.05), QRCost(base.Q, base.R, Q_terminal=base.Qt, x_goal=xg, u_goal=ug), T)
    xs, us = s.update_u(x0, us0, n_iterations=n_it)
    assert eq(xs, ref["xs"]) and eq(us, ref["us"]) and np.array_equal(s.iterations_, ref["iters"])
    # one launch per iteration, finished problems leave the launches (their goals must leave with them)
    seen = []
    t = iLQR(CartPoleDynamics(0.05), QRCost(base.Q, base.R, Q_terminal=base.Qt, x_goal=xg, u_goal=ug), T)
    xs2, us2 = t.update_u(x0, us0, n_iterations=n_it, on_iteration=lambda it, *a: seen.append(it))
    assert eq(us2, ref["us"]) and eq(xs2, ref["xs"]) and len(seen) == int(ref["iters"].max())
    xp, up = cartpole_paths(B, T, 54)
    refp = oracle_loop(lambda b: cart_model(T, path=(xp[b], up[b])), x0, us0, n_it)
    sp = iLQR(CartPoleDynamics(0.05), PathQRCost(base.Q, base.R, xp, up, Q_terminal=base.Qt), T)
    xs3, us3 = sp.update_u(x0, us0, n_iterations=n_it)
    assert eq(xs3, refp["xs"]) and eq(us3, refp["us"]) and eq(sp.J_, refp["J"])


def test_closed_loops_that_track_their_own_references(hip):
    """RecedingHorizonController over a batch of plants, every plant regulated to its own cart position: each
    loop must equal the oracle's closed loop for that plant alone (solve, plant step, augmentation, shifted
    seed, persistent regularisation state), control for control."""
    from ilqr_amd import QRCost, RecedingHorizonController, iLQR
    from ilqr_amd.cartpole import CartPoleDynamics
    from ilqr_amd.cartpole_gym import CartPoleContinuousEnv
    B, N, steps = 24, 15, 12
    r = np.random.RandomState(61)
    st = np.stack([r.uniform(-0.3, 0.3, B), np.zeros(B), np.pi + r.uniform(-0.2, 0.2, B), np.zeros(B)], 1)
    us = r.uniform(-1, 1, (B, N, 1))
    targets = np.zeros((B, 4))
    targets[:, 0] = r.uniform(-1.0, 1.0, B)
    targets[:, 2] = np.pi  # hanging: what this horizon and force limit can hold
    xg = co.augment_state(targets)
    Mo = co.cartpole_model(T=N, max_control=20.0, l=0.5, mpc=True)
    dyn = CartPoleDynamics(0.05, max_control=20.0, l=0.5)
    ctrl = iLQR(dyn, QRCost(Mo.Q, Mo.R, Q_terminal=Mo.Qt, x_goal=xg), N)

    def aug(s):
        sn, cs, _ = co.detmath(s[:, 2], s[:, 2])
        return np.stack([s[:, 0], s[:, 1], sn, cs, s[:, 3]], axis=1)
    mpc = RecedingHorizonController(aug(st), us, ctrl)
    applied, plant = [], st.copy()
    gen = mpc.control(initial_n_iterations=50, subsequent_n_iterations=5)
    for _ in range(steps):
        xs_, us_ = next(gen)
        applied.append(us_[:, 0, 0].copy())
        plant = co.plant_step(plant, us_[:, 0, 0])
        mpc.set_state(aug(plant))
    applied = np.array(applied)
    for b in range(0, B, 5):
        Mb = co.Model(co.MODEL_CARTPOLE, 5, 1, N, co.COST_QR, co.MATH_DET, True, l=0.5, Q=Mo.Q, R=Mo.R, Q_terminal=Mo.Qt,
                      x_goal=xg[b], lower=Mo.lower, upper=Mo.upper)
        s_b, u_b, delta, k, K, n_it = st[b:b + 1].copy(), us[b:b + 1].copy(), None, None, None, 50
        for step in range(steps):
            o = co.solve(Mb, co.Opts(max_iter=n_it), aug(s_b), u_b, delta=delta, k=k, K=K, trace=False)
            assert o["us"][0, 0, 0] == applied[step, b], (b, step)
            delta, k, K = o["delta"], o["k"], o["K"]
            s_b = co.plant_step(s_b, o["us"][:, 0, 0])
            u_b = np.concatenate([o["us"][:, 1:], o["us"][:, -1:]], axis=1)
            n_it = 5
    assert len(np.unique(np.round(applied[-1], 6))) > B // 2  # the plants are really doing different things
    del CartPoleContinuousEnv


def large_models(T):
    Mq = co.synthetic_model(T=T)
    kw = dict(dt=Mq.dt, A=Mq.A, B=Mq.B, W=Mq.W, sig_c=Mq.sig_c, Q=Mq.Q, R=Mq.R, Q_terminal=Mq.Qt, lower=Mq.lower,
              upper=Mq.upper)
    return Mq, kw


def test_large_state_path_with_goals_and_paths_per_problem(hip):
    T, B, n_it = 30, 14, 12
    Mq, kw = large_models(T)
    x0, us0 = co.synthetic_batch(B, T=T, seed_x=71, seed_u=72)
    r = np.random.RandomState(73)
    xg, ug = r.uniform(-0.4, 0.4, (B, 32)), r.uniform(-0.3, 0.3, (B, 8))
    ref = oracle_loop(lambda b: co.Model(co.MODEL_SIGMOID, 32, 8, T, co.COST_QR, co.MATH_DET, True, x_goal=xg[b],
                                         u_goal=ug[b], **kw), x0, us0, n_it)
    M = hip.LargeModel(T, Mq.A, Mq.B, Mq.W, Mq.Q, Mq.R, Mq.Qt, xg, ug, Mq.lower, Mq.upper, dt=Mq.dt, sig_c=Mq.sig_c)
    compare(hip.large_solve(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0)), ref, "large, goals")
    # stage entry points see the same goals
    xs_g, J_g = hip.large_rollout_cost(M, dev(x0), dev(us0))
    J_ref = np.array([co.rollout_cost(co.Model(co.MODEL_SIGMOID, 32, 8, T, co.COST_QR, co.MATH_DET, True, x_goal=xg[b],
                                               u_goal=ug[b], **kw), x0[b:b + 1], us0[b:b + 1])[1][0] for b in range(B)])
    assert eq(J_g.cpu().numpy(), J_ref)
    xp = 0.3 * np.cumsum(r.uniform(-0.1, 0.1, (B, T + 1, 32)), axis=1)
    up = r.uniform(-0.2, 0.2, (B, T, 8))
    refp = oracle_loop(lambda b: co.Model(co.MODEL_SIGMOID, 32, 8, T, co.COST_PATHQR, co.MATH_DET, True, x_goal=xp[b],
                                          u_goal=up[b], **kw), x0, us0, n_it)
    Mp = hip.LargeModel(T, Mq.A, Mq.B, Mq.W, Mq.Q, Mq.R, Mq.Qt, None, None, Mq.lower, Mq.upper, dt=Mq.dt,
                        sig_c=Mq.sig_c, x_path=xp, u_path=up)
    compare(hip.large_solve(Mp, hip.Opts(max_iter=n_it), dev(x0), dev(us0)), refp, "large, paths")
    idx = torch.tensor([1, 4, 13], device=DEV)
    part = hip.large_solve(Mp.subset(idx), hip.Opts(max_iter=n_it), dev(x0)[idx], dev(us0)[idx])
    assert eq(part["us"].cpu().numpy(), refp["us"][[1, 4, 13]])


def test_user_models_with_goals_and_paths_per_problem(hip):
    """The per-model libraries (one control: pendulum; two controls: planar rotor) with one goal, resp. one
    path, per problem, against loops of single-problem oracle solves on the same model source; enough
    iterations for searches past the first step size (evaluated by other lanes)."""
    from ilqr_amd import user_model as um
    from test_user_model_cpu import pendulum, pendulum_batch, rotor, rotor_batch
    T, B, n_it = 30, 90, 25
    Mo = pendulum(T)
    x0, us0 = pendulum_batch(B, T, seed=81)
    r = np.random.RandomState(82)
    th = r.uniform(-0.6, 0.6, B)
    xg = np.stack([np.sin(th), np.cos(th), np.zeros(B)], 1)
    ug = r.uniform(-0.2, 0.2, (B, 1))
    ref = oracle_loop(lambda b: co.user_model(Mo.user, Mo.user_params, T, Mo.Q, Mo.R, Mo.Qt, xg[b], ug[b], Mo.lower,
                                              Mo.upper), x0, us0, n_it)
    M = um.UserModel(um.load("pendulum_user"), T, Mo.user_params, Mo.Q, Mo.R, Mo.Qt, xg, ug, Mo.lower, Mo.upper)
    compare(um.solve(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0)), ref, "pendulum, goals")
    idx = torch.tensor([0, 33, 89], device=DEV)
    part = um.solve(M.subset(idx), hip.Opts(max_iter=n_it), dev(x0)[idx], dev(us0)[idx])
    assert eq(part["us"].cpu().numpy(), ref["us"][[0, 33, 89]])

    Mr = rotor(T)
    x0, us0 = rotor_batch(B, T, seed=83)
    xp = np.tile(np.asarray(Mr.x_goal)[None, None], (B, T + 1, 1))
    xp[:, :, 0] += np.cumsum(r.uniform(-0.05, 0.05, (B, T + 1)), axis=1)
    xp[:, :, 1] += r.uniform(-0.5, 0.5, (B, 1))
    up = r.uniform(-0.1, 0.1, (B, T, 2))
    refp = oracle_loop(lambda b: co.user_model(Mr.user, Mr.user_params, T, Mr.Q, Mr.R, Mr.Qt, xp[b], up[b], Mr.lower,
                                               Mr.upper, cost_kind=co.COST_PATHQR), x0, us0, n_it)
    Mp = um.UserModel(um.load("planar_rotor_user"), T, Mr.user_params, Mr.Q, Mr.R, Mr.Qt, None, None, Mr.lower, Mr.upper,
                      x_path=xp, u_path=up)
    compare(um.solve(Mp, hip.Opts(max_iter=n_it), dev(x0), dev(us0)), refp, "rotor, paths")
