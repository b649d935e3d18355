"""Generate golden vectors by RUNNING the reference's own solver code.

Run in the build container only (it reads /root/reference, which does not
travel):  ``python tests/golden/make_reference_vectors.py``.
Writes ``tests/golden/ref_*.npz``.  Nothing of the reference's text is stored:
only inputs and the numbers its functions returned.

How the reference is run here.  ``ilqr/ilqr_core.py``, ``ilqr/ilqr_boxQP.py`` and
``ilqr/ilqr_cost.py`` are Python-2 sources; they are read as text, given the
mechanical Python-3 edits listed in ``_PY3_EDITS`` IN MEMORY, and exec'd into
fresh module objects.  ``autograd.numpy`` (imported by ilqr_cost.py:8 but only
used as plain numpy by QRCost) is aliased to numpy.  The Theano-built
``CartPoleDynamics`` cannot run (no Theano), so the solver is driven with a
stand-in exposing the same interface (f, f_x, f_u, bounds, dimensions):

  * ``SympyCartPole``: the expression of ilqr/cartpole.py:72-92 typed into sympy,
    differentiated symbolically (what T.grad does) and lambdified with numpy.
  * ``OracleCartPole``: f, f_x, f_u evaluated by this repo's C oracle in libm
    mode, so that the reference's solver logic runs on bit-identical dynamics
    and can be compared tightly with the oracle's solver logic.
"""
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/ilqr"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

_PY3_EDITS = {
    "ilqr_core.py": [
        ("print Q_uu", "print(Q_uu)"),
        ("print J_opt_new - J_opt", "pass"),
        ("print J_opt_new", "pass"),
        ("print J_opt", "pass"),
        # the keyword the MPC driver passes does not exist on update_u (typo in :241)
        ("n_iterations=n_iterations, *args", "n_interations=n_iterations, *args"),
    ],
    "ilqr_boxQP.py": [("kwargs.iteritems()", "kwargs.items()")],
    "ilqr_cost.py": [],
}


def load_reference():
    """exec the three reference modules (patched in memory) and return them."""
    if "autograd" not in sys.modules:
        ag = types.ModuleType("autograd")
        ag.numpy = np
        ag.grad = ag.jacobian = None
        sys.modules["autograd"] = ag
        sys.modules["autograd.numpy"] = np
    mods = {}
    for fname in ("ilqr_boxQP.py", "ilqr_cost.py", "ilqr_core.py"):
        with open(os.path.join(REF, fname)) as fh:
            src = fh.read()
        for old, new in _PY3_EDITS[fname]:
            assert old in src, (fname, old)
            src = src.replace(old, new)
        name = fname[:-3]
        mod = types.ModuleType(name)
        mod.__file__ = os.path.join(REF, fname)
        sys.modules[name] = mod
        exec(compile(src, mod.__file__, "exec"), mod.__dict__)
        mods[name] = mod
    return mods


class SympyCartPole:
    """Stand-in for CartPoleDynamics (ilqr/cartpole.py) with sympy in Theano's place."""

    def __init__(self, dt, max_control=5, mc=1.0, mp=0.1, l=1.0, g=9.8):
        import sympy as sp
        x, x_dot, sin_theta, cos_theta, theta_dot, u = sp.symbols(
            "x x_dot sin_theta cos_theta theta_dot u", real=True)
        F = u
        temp = (F + mp * l * theta_dot ** 2 * sin_theta) / (mc + mp)
        numerator = g * sin_theta - cos_theta * temp
        denominator = l * (4.0 / 3.0 - mp * cos_theta ** 2 / (mc + mp))
        theta_dot_dot = numerator / denominator
        x_dot_dot = temp - mp * l * theta_dot_dot * cos_theta / (mc + mp)
        theta = sp.atan2(sin_theta, cos_theta)
        next_theta = theta + theta_dot * dt
        f = sp.Matrix([x + x_dot * dt, x_dot + x_dot_dot * dt, sp.sin(next_theta), sp.cos(next_theta),
                       theta_dot + theta_dot_dot * dt])
        xin = [x, x_dot, sin_theta, cos_theta, theta_dot]
        args = xin + [u]
        self._f = sp.lambdify(args, f, "numpy")
        self._fx = sp.lambdify(args, f.jacobian(xin), "numpy")
        self._fu = sp.lambdify(args, f.jacobian([u]), "numpy")
        self.upper_bound = np.array([max_control], dtype=float)
        self.lower_bound = np.array([max_control * -1], dtype=float)
        self.state_dimension = 5
        self.control_dimension = 1
        self.has_hessians = False

    def f(self, x, u, i):
        return np.asarray(self._f(*np.hstack([x, u])), dtype=float).reshape(5)

    def f_x(self, x, u, i):
        return np.asarray(self._fx(*np.hstack([x, u])), dtype=float).reshape(5, 5)

    def f_u(self, x, u, i):
        return np.asarray(self._fu(*np.hstack([x, u])), dtype=float).reshape(5, 1)


class OracleCartPole:
    """Same interface; numbers from the C oracle (libm mode)."""

    def __init__(self, dt, max_control=5, mc=1.0, mp=0.1, l=1.0, g=9.8):
        from oracle import c_oracle as co
        self._co = co
        self._M = co.Model(co.MODEL_CARTPOLE, 5, 1, 1, co.COST_QR, co.MATH_LIBM, True, dt=dt, mc=mc, mp=mp,
                           l=l, g=g, Q=np.eye(5), R=np.eye(1), lower=[-max_control], upper=[max_control])
        self.upper_bound = np.array([max_control], dtype=float)
        self.lower_bound = np.array([max_control * -1], dtype=float)
        self.state_dimension = 5
        self.control_dimension = 1
        self.has_hessians = False

    def f(self, x, u, i):
        return self._co.dynamics(self._M, x, u)[0][0]

    def f_x(self, x, u, i):
        return self._co.dynamics(self._M, x, u)[1][0]

    def f_u(self, x, u, i):
        return self._co.dynamics(self._M, x, u)[2][0]


def augment(st):
    st = np.asarray(st, dtype=float)
    return np.stack([st[..., 0], st[..., 1], np.sin(st[..., 2]), np.cos(st[..., 2]), st[..., 3]], axis=-1)


def example_cost(mods, mpc=False):
    """examples/cartpole_example.py:15-27 / examples/cartpole_example_mpc.py:84-95"""
    Q = 100 * np.eye(5)
    if mpc:
        Q[0, 0] = Q[1, 1] = 10.0
        Q[4, 4] = 80
        R = np.array([[1.0]])
    else:
        Q[1, 1] = Q[4, 4] = 10.0
        R = np.array([[5.0]])
    x_goal = augment(np.array([0.0, 0.0, 0.0, 0.0]))
    return mods["ilqr_cost"].QRCost(Q, R, Q_terminal=100 * np.eye(5), x_goal=x_goal), Q, R


def run_update_u(mods, dynamics, cost, N, x0, us_init, **kw):
    """One reference update_u call with everything observable recorded.

    Beyond what the on_iteration callback hands out (ilqr_core.py:342-345), three of the solver's own calls
    are watched from outside, without touching its code: every ``boxQP`` call of ``_backward_pass`` (:156:
    the result code and the free mask it returns per time step), the Delta_J1 / Delta_J2 that
    ``_backward_pass`` returns (:289), and every ``_trajectory_cost`` of the line search (:303), from which
    the ratio tested at :310 and the improvement tested at :327 are recomputed -- the margins by which the
    reference's discrete decisions were taken."""
    core = mods["ilqr_core"]
    solver = core.iLQR(dynamics, cost, N)
    m = dynamics.control_dimension
    scale = kw.get("scale", 0.1)
    tol = kw.get("tolerance", 1e-4)
    alphas = 1.1 ** (-np.arange(kw.get("line_search_steps", 30)) ** 2)
    rec = dict(J=[], accepted=[], converged=[], mu=[], delta=[], n_forward=[], qp_result=[], free=[], dJ1=[],
               dJ2=[], ratio=[], ratio_margin=[], actual=[])
    calls = [0]
    orig_forward = solver._forward_pass

    def counting_forward(*a, **k):
        calls[0] += 1
        return orig_forward(*a, **k)

    solver._forward_pass = counting_forward

    qp_calls, last_dJ, costs = [], [0.0, 0.0], []
    orig_boxqp, orig_backward, orig_cost = core.boxQP, solver._backward_pass, solver._trajectory_cost

    def recording_boxqp(H, g, lower, upper, x0_, **opts):
        out = orig_boxqp(H, g, lower, upper, x0_, **opts)
        qp_calls.append((int(out[1]), np.array(out[3], dtype=bool).reshape(m).copy()))
        return out

    def recording_backward(us, xs):
        del qp_calls[:]  # a retry after LinAlgError starts the sweep again (:287-299)
        out = orig_backward(us, xs)
        last_dJ[0], last_dJ[1] = float(out[2]), float(out[3])
        return out

    def recording_cost(xs, us):
        J = orig_cost(xs, us)
        costs.append(float(J))
        return J

    core.boxQP = recording_boxqp
    solver._backward_pass = recording_backward
    solver._trajectory_cost = recording_cost
    J_prev = [None]

    def on_iteration(it, xs, us, J, accepted, converged):
        if J_prev[0] is None:  # the first recorded cost is that of the initial rollout (:271)
            J_prev[0] = costs.pop(0)
        rec["J"].append(float(J))
        rec["accepted"].append(bool(accepted))
        rec["converged"].append(bool(converged))
        rec["mu"].append(float(solver.mu_))
        rec["delta"].append(float(solver.delta_))
        rec["n_forward"].append(calls[0])
        calls[0] = 0
        # boxQP is called for i = N-1 ... 0 (:127); a non-PD break (:137-140) leaves the earlier steps unvisited
        qr = np.full(N, -100, dtype=np.int32)
        fr = np.zeros((N, m), dtype=bool)
        for c, (result, free) in enumerate(qp_calls):
            qr[N - 1 - c], fr[N - 1 - c] = result, free
        rec["qp_result"].append(qr)
        rec["free"].append(fr)
        rec["dJ1"].append(last_dJ[0])
        rec["dJ2"].append(last_dJ[1])
        # the ratios the line search compared with `scale` (:305-310), one per cost it evaluated
        ratios = []
        for a, Jn in zip(alphas, costs):
            expected = (a * last_dJ[0] + a ** 2 * last_dJ[1]) * -1
            if expected <= 0:
                break
            ratios.append((J_prev[0] - Jn) / expected)
        rec["ratio"].append(ratios[-1] if ratios else np.nan)
        finite = [abs(r - scale) for r in ratios if np.isfinite(r)]
        rec["ratio_margin"].append(min(finite) if finite else np.nan)
        rec["actual"].append(J_prev[0] - costs[-1] if accepted else np.nan)
        del costs[:]
        J_prev[0] = float(J)

    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xs, us = solver.update_u(x0, us_init, on_iteration=on_iteration, **kw)
    finally:
        core.boxQP = orig_boxqp
    out = dict(x0=x0, us_init=us_init, xs=np.asarray(xs), us=np.asarray(us), k=np.asarray(solver.k_),
               K=np.asarray(solver.K_), mu_final=float(solver.mu_), delta_final=float(solver.delta_),
               scale=scale, tolerance=tol)
    for key, val in rec.items():
        out["it_" + key] = np.asarray(val)
    return out


def pack(prefix, d, store):
    for key, val in d.items():
        store[prefix + key] = val


def make_solver_vectors(mods):
    store = {}
    x0_c0 = augment(np.array([0.0, 0.0, np.pi, 0.0]))
    rs = np.random.RandomState(1)
    B = 6
    st = np.stack([rs.uniform(-1, 1, B), rs.uniform(-1, 1, B), rs.uniform(-np.pi, np.pi, B),
                   rs.uniform(-1, 1, B)], axis=1)
    x0s = [x0_c0] + list(augment(st))
    us_all = np.random.RandomState(2).uniform(-1, 1, (B, 100, 1))
    uss = [np.random.RandomState(0).uniform(-1, 1, (100, 1))] + list(us_all)
    for tag, Dyn in (("sympy", SympyCartPole), ("oracle", OracleCartPole)):
        dyn = Dyn(0.05)
        cost, Q, R = example_cost(mods)
        for i, (x0, us) in enumerate(zip(x0s, uss)):
            n_it = 200 if i < 3 else 60
            out = run_update_u(mods, dyn, cost, 100, x0, us, n_interations=n_it)
            out["max_iter"] = n_it
            pack("%s_p%d_" % (tag, i), out, store)
            print(tag, i, "iters", len(out["it_J"]), "J", out["it_J"][-1], "conv", out["it_converged"][-1])
    store["n_problems"] = len(x0s)
    np.savez_compressed(os.path.join(HERE, "ref_cartpole_solve.npz"), **store)


def make_boxqp_vectors(mods):
    boxQP = mods["ilqr_boxQP"].boxQP
    rs = np.random.RandomState(7)
    store = {}
    idx = 0
    for m in (1, 1, 1, 2, 2, 3, 6, 6, 8):
        for rep in range(6):
            A = rs.randn(m, m)
            H = A.T.dot(A) + 0.1 * np.eye(m)
            g = rs.randn(m) * 3
            width = [10.0, 1.0, 0.3][rep % 3]
            lower = -np.ones(m) * width * rs.uniform(0.5, 1.0, m)
            upper = np.ones(m) * width * rs.uniform(0.5, 1.0, m)
            x0 = rs.randn(m) if rep % 2 == 0 else None
            step_dec = 0.9 if rep < 3 else 0.6
            kw = {'maxIter': 100, 'minGrad': 1e-8, 'minRelImprove': 1e-8, 'stepDec': step_dec,
                  'minStep': 1e-22, 'Armijo': 0.1, 'verbose': 0}
            x, result, Hff, free = boxQP(H, g, lower, upper, x0, **kw)
            p = "q%d_" % idx
            store[p + "H"], store[p + "g"], store[p + "lower"], store[p + "upper"] = H, g, lower, upper
            store[p + "x0"] = np.full(m, np.nan) if x0 is None else x0
            store[p + "has_x0"] = x0 is not None
            store[p + "step_dec"] = step_dec
            store[p + "x"], store[p + "result"], store[p + "Hff"], store[p + "free"] = x, result, Hff, free
            idx += 1
    # a non-PD Hessian, an all-clamped start and the 1-D un-factorised first step
    for H, g, lo, up, x0 in (
            (np.array([[1.0, 2.0], [2.0, 1.0]]), np.array([1.0, -1.0]), -np.ones(2), np.ones(2), np.zeros(2)),
            (np.eye(2), np.array([5.0, -5.0]), -np.ones(2), np.ones(2), np.array([-3.0, 3.0])),
            (np.array([[4.0]]), np.array([-2.0]), np.array([-5.0]), np.array([5.0]), np.array([0.25])),
            (np.array([[4.0]]), np.array([-2.0]), np.array([-5.0]), np.array([5.0]), np.array([0.5])),
            (np.array([[30.0]]), np.array([-400.0]), np.array([-4.0]), np.array([6.0]), np.array([13.3]))):
        kw = {'maxIter': 100, 'minGrad': 1e-8, 'minRelImprove': 1e-8, 'stepDec': 0.9,
              'minStep': 1e-22, 'Armijo': 0.1, 'verbose': 0}
        x, result, Hff, free = boxQP(H, g, lo, up, x0, **kw)
        p = "q%d_" % idx
        store[p + "H"], store[p + "g"], store[p + "lower"], store[p + "upper"] = H, g, lo, up
        store[p + "x0"], store[p + "has_x0"], store[p + "step_dec"] = x0, True, 0.9
        store[p + "x"], store[p + "result"], store[p + "Hff"], store[p + "free"] = x, result, Hff, free
        idx += 1
    # hunt for the rarer exits: 2 (line search exhausted) and 4 (small improvement)
    want = {2: 5, 4: 5}
    rs = np.random.RandomState(19)
    tries = 0
    while any(v > 0 for v in want.values()) and tries < 20000:
        tries += 1
        m = int(rs.randint(1, 4))
        A = rs.randn(m, m)
        H = A.T.dot(A) * rs.uniform(1, 50) + 0.1 * np.eye(m)
        g = rs.randn(m) * 3
        lower = -rs.uniform(0.05, 1.0, m)
        upper = rs.uniform(0.05, 1.0, m)
        x0 = np.where(rs.uniform(size=m) < 0.5, upper, rs.randn(m))
        kw = {'maxIter': 100, 'minGrad': 1e-8, 'minRelImprove': 1e-8, 'stepDec': 0.9,
              'minStep': 1e-22, 'Armijo': 0.1, 'verbose': 0}
        x, result, Hff, free = boxQP(H, g, lower, upper, x0, **kw)
        if want.get(int(result), 0) > 0:
            want[int(result)] -= 1
            p = "q%d_" % idx
            store[p + "H"], store[p + "g"], store[p + "lower"], store[p + "upper"] = H, g, lower, upper
            store[p + "x0"], store[p + "has_x0"], store[p + "step_dec"] = x0, True, 0.9
            store[p + "x"], store[p + "result"], store[p + "Hff"], store[p + "free"] = x, result, Hff, free
            idx += 1
    print("rare exits still missing:", want)
    store["n_cases"] = idx
    np.savez_compressed(os.path.join(HERE, "ref_boxqp.npz"), **store)
    print("boxqp cases", idx)


def make_stage_vectors(mods):
    """Single calls of the reference's rollout / cost / backward / forward."""
    dyn = OracleCartPole(0.05)
    cost, Q, R = example_cost(mods)
    solver = mods["ilqr_core"].iLQR(dyn, cost, 100)
    rs = np.random.RandomState(11)
    store = {}
    for i in range(3):
        st = np.array([rs.uniform(-1, 1), rs.uniform(-1, 1), rs.uniform(-np.pi, np.pi), rs.uniform(-1, 1)])
        x0 = augment(st)
        us = rs.uniform(-6, 6, (100, 1)) if i == 2 else rs.uniform(-1, 1, (100, 1))
        xs = solver._dynamics_rollout(x0, us)
        J = solver._trajectory_cost(xs, us)
        solver.mu_ = [1.0, 0.0, 37.5][i]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            k, K, d1, d2 = solver._backward_pass(us, xs)
        alphas = 1.1 ** (-np.arange(4) ** 2)
        fw = [solver._forward_pass(xs, us, k, K, a) for a in alphas]
        p = "s%d_" % i
        store[p + "x0"], store[p + "us"], store[p + "xs"], store[p + "J"] = x0, us, xs, J
        store[p + "mu"], store[p + "k"], store[p + "K"] = solver.mu_, k, K
        store[p + "dJ1"], store[p + "dJ2"] = d1, d2
        store[p + "alphas"] = alphas
        store[p + "xs_new"] = np.stack([f[0] for f in fw])
        store[p + "us_new"] = np.stack([f[1] for f in fw])
        store[p + "J_new"] = np.array([solver._trajectory_cost(f[0], f[1]) for f in fw])
    store["n_cases"] = 3
    np.savez_compressed(os.path.join(HERE, "ref_stages.npz"), **store)
    print("stage cases 3")


def make_mpc_vectors(mods):
    """RecedingHorizonController.control (ilqr_core.py:422-466) closed on the
    plant of ilqr/cartpole_gym.py:60-72.  gym is absent, so the plant step is
    this repo's restatement of _state_eq (C oracle, libm mode) and the loop
    feeds it back through set_state(), as examples/cartpole_example_mpc.py:74-75 does."""
    from oracle import c_oracle as co
    dyn = OracleCartPole(0.05, max_control=20.0, l=0.5)
    cost, Q, R = example_cost(mods, mpc=True)
    N = 15
    solver = mods["ilqr_core"].iLQR(dyn, cost, N)
    st = np.array([0.0, 0.0, 0.5 * np.pi, 0.0])
    us_init = np.random.RandomState(3).uniform(-1, 1, (N, 1))
    ctrl = mods["ilqr_core"].RecedingHorizonController(augment(st), us_init, solver, None)
    steps = 40
    u_applied, states, mus, deltas = [], [st.copy()], [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gen = ctrl.control(step_size=1, initial_n_iterations=200, subsequent_n_iterations=10)
        for _ in range(steps):
            xs, us = next(gen)
            u0 = float(us[0][0])
            st = co.plant_step(st, [u0], math_mode=co.MATH_LIBM)[0]
            ctrl.set_state(augment(st))
            u_applied.append(u0)
            states.append(st.copy())
            mus.append(solver.mu_)
            deltas.append(solver.delta_)
    np.savez_compressed(os.path.join(HERE, "ref_mpc.npz"), st0=states[0], us_init=us_init, N=N, steps=steps,
                        u_applied=np.array(u_applied), states=np.array(states), mu=np.array(mus),
                        delta=np.array(deltas))
    print("mpc final state", states[-1])


def pathqr_problem(N=60):
    """A trajectory-following problem for PathQRCost (ilqr_cost.py:635-770): track a smooth
    swing from hanging to upright while the cart moves out and back."""
    t = np.linspace(0.0, 1.0, N + 1)
    theta = np.pi * (1.0 - t) ** 2
    st = np.stack([0.4 * np.sin(np.pi * t), 0.4 * np.pi * np.cos(np.pi * t) / (N * 0.05),
                   theta, np.gradient(theta, 0.05 * N / N)], axis=1)
    x_path = augment(st)
    u_path = 0.5 * np.cos(3.0 * np.pi * t[:-1]).reshape(N, 1)
    Q = np.diag([20.0, 2.0, 50.0, 50.0, 5.0])
    Q[0, 1] = 1.5                      # a non-symmetric Q: l_x uses Q + Q^T, l uses Q
    R = np.array([[0.8]])
    Qt = 80.0 * np.eye(5)
    return x_path, u_path, Q, R, Qt


def make_pathqr_vectors(mods):
    """update_u of the reference with its PathQRCost, dynamics from the oracle (libm mode)."""
    N = 60
    x_path, u_path, Q, R, Qt = pathqr_problem(N)
    cost = mods["ilqr_cost"].PathQRCost(Q, R, x_path, u_path, Q_terminal=Qt)
    dyn = OracleCartPole(0.05)
    rs = np.random.RandomState(23)
    store = dict(N=N, x_path=x_path, u_path=u_path, Q=Q, R=R, Qt=Qt)
    for i in range(3):
        st = np.array([rs.uniform(-0.3, 0.3), rs.uniform(-0.3, 0.3), np.pi + rs.uniform(-0.4, 0.4),
                       rs.uniform(-0.5, 0.5)])
        us = rs.uniform(-1, 1, (N, 1))
        out = run_update_u(mods, dyn, cost, N, augment(st), us, n_interations=50)
        out["max_iter"] = 50
        # single calls of the cost itself
        out["l0"] = cost.l(out["xs"][3], out["us"][3], 3)
        out["lT"] = cost.l(out["xs"][N], None, N, terminal=True)
        out["lx0"] = cost.l_x(out["xs"][3], out["us"][3], 3)
        out["lu0"] = cost.l_u(out["xs"][3], out["us"][3], 3)
        pack("p%d_" % i, out, store)
        print("pathqr", i, "iters", len(out["it_J"]), "J", out["it_J"][-1], "conv", out["it_converged"][-1])
    store["n_problems"] = 3
    np.savez_compressed(os.path.join(HERE, "ref_pathqr_solve.npz"), **store)


if __name__ == "__main__":
    mods = load_reference()
    which = sys.argv[1:] or ["boxqp", "stages", "solve", "mpc", "pathqr"]
    if "pathqr" in which:
        make_pathqr_vectors(mods)
    if "boxqp" in which:
        make_boxqp_vectors(mods)
    if "stages" in which:
        make_stage_vectors(mods)
    if "solve" in which:
        make_solver_vectors(mods)
    if "mpc" in which:
        make_mpc_vectors(mods)
