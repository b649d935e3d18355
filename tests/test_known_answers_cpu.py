"""Known-answer checks of the checker itself (CPU): results that come from neither the oracle's author
nor the reference -- a linear-quadratic problem solved with plain numpy, and torch's autograd on the
synthetic model's expression."""
import numpy as np
import pytest

from oracle import c_oracle as co

import lq_known_answer as lq


def lq_oracle_model(p, math_mode=co.MATH_DET):
    # the synthetic model x+ = x + dt (A x + B u + c phi(W x)) with c = 0 is the linear model
    big = 1e6
    return co.Model(co.MODEL_SIGMOID, p["n"], p["m"], p["T"], co.COST_PATHQR, math_mode, True, dt=p["dt"], A=p["A"],
                    B=p["B"], W=np.zeros((p["n"], p["n"])), sig_c=0.0, Q=p["Q"], R=p["R"], Q_terminal=p["Qf"],
                    x_goal=p["g"], u_goal=p["h"], lower=np.full(p["m"], -big), upper=np.full(p["m"], big))


@pytest.mark.parametrize("T,seed", [(25, 17), (60, 18)])
def test_backward_pass_is_the_riccati_recursion_and_one_full_step_is_optimal(T, seed):
    p = lq.problem(T=T, seed=seed)
    Mo = lq_oracle_model(p)
    xs, J0 = co.rollout_cost(Mo, p["x0"], p["us_start"])
    np.testing.assert_allclose(xs[0], lq.rollout(p, p["us_start"]), rtol=0, atol=1e-12)
    assert abs(J0[0] - lq.cost(p, xs[0], p["us_start"])) <= 1e-12 * J0[0]
    bw = co.backward(Mo, co.Opts(), xs, p["us_start"], 0.0)
    assert bw["stopped_at"][0] == -1 and bw["n_clamped"][0] == 0 and (bw["qp_result"] == 5).all()
    xs_new, us_new, J_new = co.forward(Mo, [1.0], xs, p["us_start"], bw["k"], bw["K"])
    lq.check(p, bw["k"][0], bw["K"][0], bw["dJ1"][0], bw["dJ2"][0], us_new[0, 0], J_new[0, 0])
    # and the whole solver: from mu = 1 it needs a few iterations, but it must end on the same optimum
    out = co.solve(Mo, co.Opts(max_iter=50, tol=1e-12), p["x0"], p["us_start"], trace=False)
    _, _, J_opt = lq.dense_optimum(p)
    assert abs(out["J"][0] - J_opt) <= 1e-9 * J_opt


def test_synthetic_model_jacobian_matches_autograd():
    """f_x, f_u of x + dt (A x + B u + c phi(W x)), phi(z) = z / sqrt(1 + z^2), as the oracle computes them
    (the formula the device code restates) against torch's fp64 autograd of that expression."""
    torch = pytest.importorskip("torch")
    Mo = co.synthetic_model(T=4)
    A, B, W = (torch.tensor(a, dtype=torch.float64) for a in (Mo.A, Mo.B, Mo.W))
    dt, c = Mo.dt, Mo.sig_c

    def f(x, u):
        z = W @ x
        return x + dt * (A @ x + B @ u + c * z / torch.sqrt(1.0 + z * z))
    r = np.random.RandomState(3)
    n_pts = 240
    xs = np.concatenate([r.uniform(-1, 1, (n_pts // 2, 32)), r.uniform(-8, 8, (n_pts // 2, 32))])
    us = r.uniform(-1, 1, (n_pts, 8))
    fo, fxo, fuo = co.dynamics(Mo, xs, us)
    for i in range(n_pts):
        x, u = torch.tensor(xs[i]), torch.tensor(us[i])
        Jx, Ju = torch.autograd.functional.jacobian(f, (x, u))
        np.testing.assert_allclose(fo[i], f(x, u).numpy(), rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(fxo[i], Jx.numpy(), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(fuo[i], Ju.numpy(), rtol=1e-12, atol=1e-13)
