"""CPU tests: the C oracle against numbers recorded from the reference's own code.

The fixtures in tests/golden/ref_*.npz were produced by
tests/golden/make_reference_vectors.py, which exec's ilqr_core.py / ilqr_boxQP.py /
ilqr_cost.py of the reference (with mechanical Python-3 edits made in memory) in
the build container.  The oracle orders floating-point sums its own way
(left to right, no BLAS), so values are compared with a tolerance and every
discrete decision (accept flags, number of line-search rollouts, mu schedule,
active sets, result codes) must be identical.
"""
import numpy as np
import pytest

from oracle import c_oracle as co

RTOL_J = 1e-9  # relative tolerance on costs, the bar named for this solver


@pytest.fixture(scope="module")
def solve_vec(golden_dir):
    return np.load(golden_dir + "/ref_cartpole_solve.npz")


def _n_forward(alpha_idx, n_alpha=30):
    return np.where(alpha_idx < n_alpha, alpha_idx + 1, n_alpha)


@pytest.mark.parametrize("tag", ["oracle", "sympy"])
@pytest.mark.parametrize("math_mode", [co.MATH_LIBM, co.MATH_DET])
def test_update_u_matches_reference_run(solve_vec, tag, math_mode):
    g = solve_vec
    M = co.cartpole_model(math_mode=math_mode)
    for i in range(int(g["n_problems"])):
        p = "%s_p%d_" % (tag, i)
        out = co.solve(M, co.Opts(max_iter=int(g[p + "max_iter"])), g[p + "x0"], g[p + "us_init"])
        tr, it = out["trace"], int(out["iters"][0])
        ref_J = g[p + "it_J"]
        assert it == len(ref_J), (tag, i)
        np.testing.assert_allclose(tr["J"][0, :it], ref_J, rtol=RTOL_J)
        assert np.array_equal(tr["accepted"][0, :it].astype(bool), g[p + "it_accepted"])
        assert np.array_equal(_n_forward(tr["alpha_idx"][0, :it]), g[p + "it_n_forward"])
        assert np.array_equal(tr["mu"][0, :it], g[p + "it_mu"])
        assert np.array_equal(tr["delta"][0, :it], g[p + "it_delta"])
        assert bool(out["converged"][0]) == bool(g[p + "it_converged"][-1])
        assert out["mu"][0] == g[p + "mu_final"] and out["delta"][0] == g[p + "delta_final"]
        np.testing.assert_allclose(out["xs"][0], g[p + "xs"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(out["us"][0], g[p + "us"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(out["k"][0], g[p + "k"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(out["K"][0], g[p + "K"], rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("tag", ["oracle", "sympy"])
def test_boxqp_outcomes_per_step_and_decision_margins_match_reference_run(solve_vec, tag):
    """What the reference's boxQP returned at EVERY time step of EVERY backward pass (ilqr_core.py:156: result
    code and free mask, recorded by watching the call from outside) against the oracle's, plus the two
    quantities its discrete decisions were taken on: the line-search ratio (:310) and the accepted
    improvement (:327).  The recorded margins show how far the run stayed from a tie."""
    g = solve_vec
    M = co.cartpole_model(math_mode=co.MATH_LIBM)
    n_codes = set()
    for i in range(int(g["n_problems"])):
        p = "%s_p%d_" % (tag, i)
        out = co.solve_with_step_history(M, co.Opts(max_iter=int(g[p + "max_iter"])), g[p + "x0"], g[p + "us_init"])
        it = int(out["iters"][0])
        assert it == len(g[p + "it_J"])
        assert np.array_equal(out["hist_qp_result"], g[p + "it_qp_result"]), (tag, i)
        assert np.array_equal(out["hist_free"], g[p + "it_free"]), (tag, i)
        reached = g[p + "it_qp_result"] != -100
        n_clamped_ref = (reached & ~g[p + "it_free"][:, :, 0]).sum(axis=1)
        assert np.array_equal(out["trace"]["n_clamped"][0, :it], n_clamped_ref)
        first_unreached = np.where(reached.all(axis=1), -1, reached.shape[1] - 1 - reached[:, ::-1].sum(axis=1))
        assert np.array_equal(out["trace"]["stopped_at"][0, :it], first_unreached)
        n_codes |= set(np.unique(g[p + "it_qp_result"]))
        # decisions: same ratio and improvement as the reference computed, and no decision near a tie
        np.testing.assert_allclose(out["trace"]["ratio"][0, :it], g[p + "it_ratio"], rtol=1e-6, equal_nan=True)
        scale, tol = float(g[p + "scale"]), float(g[p + "tolerance"])
        assert np.nanmin(g[p + "it_ratio_margin"]) > 1e-3 * scale
        acc = g[p + "it_accepted"]
        assert np.nanmin(np.abs(g[p + "it_actual"][acc] - tol)) > 1e-3 * tol
    assert {2, 5, 6} <= n_codes  # line search exhausted, small gradient, all clamped: all occur


def pathqr_model(g, math_mode):
    N = int(g["N"])
    return co.Model(co.MODEL_CARTPOLE, 5, 1, N, co.COST_PATHQR, math_mode, True, dt=0.05, Q=g["Q"], R=g["R"],
                    Q_terminal=g["Qt"], x_goal=g["x_path"], u_goal=g["u_path"], lower=[-5.0], upper=[5.0])


@pytest.mark.parametrize("math_mode", [co.MATH_LIBM, co.MATH_DET])
def test_pathqr_update_u_matches_reference_run(golden_dir, math_mode):
    """The reference's PathQRCost (ilqr_cost.py:635-770: time-indexed goals, no 1/2 in the value,
    a non-symmetric Q) driving its update_u."""
    g = np.load(golden_dir + "/ref_pathqr_solve.npz")
    M = pathqr_model(g, math_mode)
    N = int(g["N"])
    for i in range(int(g["n_problems"])):
        p = "p%d_" % i
        out = co.solve(M, co.Opts(max_iter=int(g[p + "max_iter"])), g[p + "x0"], g[p + "us_init"])
        tr, it = out["trace"], int(out["iters"][0])
        ref_J = g[p + "it_J"]
        assert it == len(ref_J), i
        np.testing.assert_allclose(tr["J"][0, :it], ref_J, rtol=RTOL_J)
        assert np.array_equal(tr["accepted"][0, :it].astype(bool), g[p + "it_accepted"])
        assert np.array_equal(_n_forward(tr["alpha_idx"][0, :it]), g[p + "it_n_forward"])
        assert np.array_equal(tr["mu"][0, :it], g[p + "it_mu"])
        assert bool(out["converged"][0]) == bool(g[p + "it_converged"][-1])
        np.testing.assert_allclose(out["xs"][0], g[p + "xs"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(out["us"][0], g[p + "us"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(out["K"][0], g[p + "K"], rtol=1e-7, atol=1e-7)
        # the cost functions themselves at one interior step and at the end
        d = co.derivatives(M, g[p + "xs"], g[p + "us"])
        np.testing.assert_allclose(d["lx"][3], g[p + "lx0"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(d["lu"][3], g[p + "lu0"], rtol=1e-12, atol=1e-12)


def test_boxqp_matches_reference(golden_dir):
    g = np.load(golden_dir + "/ref_boxqp.npz")
    n = int(g["n_cases"])
    seen = set()
    for i in range(n):
        p = "q%d_" % i
        O = co.Opts(qp_step_dec=float(g[p + "step_dec"]))
        x0 = g[p + "x0"] if bool(g[p + "has_x0"]) else None
        out = co.boxqp(g[p + "H"], g[p + "g"], g[p + "lower"], g[p + "upper"], x0, O)
        assert out["rc"][0] == 0
        assert out["result"][0] == int(g[p + "result"]), i
        assert np.array_equal(out["free"][0], g[p + "free"]), i
        np.testing.assert_allclose(out["x"][0], g[p + "x"], rtol=1e-9, atol=1e-12)
        ref_H = g[p + "Hff"]
        d = ref_H.shape[0]
        assert out["hff_dim"][0] == d, i
        # only the lower triangle is meaningful (scipy leaves the upper one untouched)
        np.testing.assert_allclose(np.tril(out["Hff"][0][:d, :d]), np.tril(ref_H), rtol=1e-9, atol=1e-12)
        seen.add(int(g[p + "result"]))
    # the fixture set exercises these exits of the reference
    assert {-1, 2, 4, 5, 6} <= seen


def test_stages_match_reference(golden_dir):
    g = np.load(golden_dir + "/ref_stages.npz")
    M = co.cartpole_model(math_mode=co.MATH_LIBM)
    for i in range(int(g["n_cases"])):
        p = "s%d_" % i
        xs, J = co.rollout_cost(M, g[p + "x0"], g[p + "us"])
        np.testing.assert_allclose(xs[0], g[p + "xs"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(J[0], g[p + "J"], rtol=1e-13)
        bw = co.backward(M, co.Opts(), g[p + "xs"], g[p + "us"], float(g[p + "mu"]))
        assert bw["rc"][0] == 0
        np.testing.assert_allclose(bw["k"][0], g[p + "k"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(bw["K"][0], g[p + "K"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(bw["dJ1"][0], g[p + "dJ1"], rtol=1e-10)
        np.testing.assert_allclose(bw["dJ2"][0], g[p + "dJ2"], rtol=1e-10)
        xs_new, us_new, J_new = co.forward(M, g[p + "alphas"], g[p + "xs"], g[p + "us"], g[p + "k"], g[p + "K"])
        np.testing.assert_allclose(xs_new[0], g[p + "xs_new"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(us_new[0], g[p + "us_new"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(J_new[0], g[p + "J_new"], rtol=1e-11)


def test_mpc_matches_reference(golden_dir):
    """Closed loop of ilqr_core.py:422-466 + the plant of cartpole_gym.py:60-72,
    including two re-solves in which Q_uu loses positive definiteness."""
    g = np.load(golden_dir + "/ref_mpc.npz")
    N, steps = int(g["N"]), int(g["steps"])
    M = co.cartpole_model(T=N, math_mode=co.MATH_LIBM, max_control=20.0, l=0.5, mpc=True)
    st = g["st0"].copy()
    us = g["us_init"].copy()
    delta = k = K = None
    n_it = 200
    for s in range(steps):
        out = co.solve(M, co.Opts(max_iter=n_it), co.augment_state(st), us, delta=delta, k=k, K=K)
        delta, k, K = out["delta"], out["k"], out["K"]
        u0 = out["us"][0, 0, 0]
        np.testing.assert_allclose(u0, g["u_applied"][s], rtol=1e-7, atol=1e-9)
        assert out["mu"][0] == g["mu"][s] and out["delta"][0] == g["delta"][s], s
        st = co.plant_step(st, [u0], math_mode=co.MATH_LIBM)[0]
        np.testing.assert_allclose(st, g["states"][s + 1], rtol=1e-7, atol=1e-9)
        # ilqr_core.py:463-465 with step_size = 1
        us = np.vstack([out["us"][0, 1:], out["us"][0, -1:]])
        n_it = 10


def test_det_math_within_one_ulp_of_libm():
    r = np.random.RandomState(0)
    a = np.concatenate([r.uniform(-10, 10, 400000), r.uniform(-1e5, 1e5, 200000), r.uniform(-4, 4, 200000),
                        r.uniform(-1e-3, 1e-3, 50000)])
    b = r.uniform(-10, 10, a.size)
    s, c, t = co.detmath(a, b)

    def ulp(x, ref):
        return np.abs(x - ref) / np.spacing(np.abs(ref))
    assert ulp(s, np.sin(a)).max() <= 1.0
    assert ulp(c, np.cos(a)).max() <= 1.0
    assert ulp(t, np.arctan2(a, b)).max() <= 1.0
    # axes, signed zeros, non-finite input
    y = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.inf, np.nan, 1.0])
    x = np.array([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, np.inf, -np.inf, 1.0, 1.0, np.nan])
    _, _, t = co.detmath(y, x)
    ref = np.arctan2(y, x)
    assert np.array_equal(np.isnan(t), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(t[ok], ref[ok], rtol=2e-16)
    assert np.array_equal(np.signbit(t[ok]), np.signbit(ref[ok]))
    s, c, _ = co.detmath(np.array([np.inf, np.nan, 2.0 ** 31]), np.ones(3))
    assert np.isnan(s).all() and np.isnan(c).all()


def test_cartpole_jacobian_matches_autograd():
    """Analytic f_x, f_u of the oracle against torch fp64 autograd of the
    expression in ilqr/cartpole.py:72-92 (sin/cos of theta as independent inputs)."""
    import torch
    dt, mc, mp, l, g = 0.05, 1.0, 0.1, 1.0, 9.8

    def f(z):
        x, xd, s, c, thd, F = z
        temp = (F + mp * l * thd ** 2 * s) / (mc + mp)
        thdd = (g * s - c * temp) / (l * (4.0 / 3.0 - mp * c ** 2 / (mc + mp)))
        xdd = temp - mp * l * thdd * c / (mc + mp)
        nth = torch.atan2(s, c) + thd * dt
        return torch.stack([x + xd * dt, xd + xdd * dt, torch.sin(nth), torch.cos(nth), thd + thdd * dt])

    r = np.random.RandomState(5)
    M = co.cartpole_model(math_mode=co.MATH_LIBM)
    pts = r.uniform(-2, 2, (200, 6))
    pts[:, 4] *= 4
    pts[:, 5] *= 3
    fo, fx, fu = co.dynamics(M, pts[:, :5], pts[:, 5:])
    for i in range(pts.shape[0]):
        z = torch.tensor(pts[i], dtype=torch.float64)
        J = torch.autograd.functional.jacobian(f, z).numpy()
        np.testing.assert_allclose(fo[i], f(z).numpy(), rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(fx[i], J[:, :5], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(fu[i], J[:, 5:], rtol=1e-11, atol=1e-12)
