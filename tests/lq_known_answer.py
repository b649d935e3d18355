"""A known answer that owes nothing to the oracle or to the reference's code: the finite-horizon
linear-quadratic tracking problem

    minimise  sum_t (x_t - g_t)' Q (x_t - g_t) + (u_t - h_t)' R (u_t - h_t)  +  (x_T - g_T)' Qf (x_T - g_T)
    subject to x_{t+1} = F x_t + G u_t,

solved here twice with plain numpy: as one dense least-squares system in all the controls (the optimum
U*, J*), and by the textbook Riccati recursion (the feedback gains).  iLQR's backward pass
(ilqr/ilqr_core.py:109-228) with mu = 0 and limits that never bind IS that recursion -- its Q-function
expansion (:206-212) is exact for a linear model and a quadratic cost -- and one forward pass with alpha = 1
(:85-107) lands on the optimum from any starting trajectory.  The cost is the reference's PathQRCost
(ilqr/ilqr_cost.py:635-770), whose value and derivatives are consistent with each other (QRCost's value
carries a 1/2 that its gradient lacks, :547 vs :569)."""
import numpy as np


def problem(n=32, m=8, T=25, dt=0.05, seed=17):
    r = np.random.RandomState(seed)
    Gm = r.randn(n, n)
    A = 0.8 * Gm / np.abs(np.linalg.eigvals(Gm)).max() - 0.5 * np.eye(n)
    B = r.randn(n, m) / np.sqrt(m)
    Q = np.diag(r.uniform(0.5, 2.0, n)) + 0.05 * r.randn(n, n)   # not symmetric: only Q + Q' matters
    Qf = 5.0 * np.eye(n) + 0.1 * r.randn(n, n)
    R = np.diag(r.uniform(0.05, 0.2, m)) + 0.005 * r.randn(m, m)
    g = 0.3 * np.cumsum(r.uniform(-0.1, 0.1, (T + 1, n)), axis=0)
    h = r.uniform(-0.2, 0.2, (T, m))
    x0 = r.uniform(-1.0, 1.0, n)
    us_start = r.uniform(-0.5, 0.5, (T, m))
    return dict(n=n, m=m, T=T, dt=dt, A=A, B=B, Q=Q, Qf=Qf, R=R, g=g, h=h, x0=x0, us_start=us_start,
                F=np.eye(n) + dt * A, G=dt * B)


def cost(p, xs, us):
    J = 0.0
    for t in range(p["T"]):
        dx, du = xs[t] - p["g"][t], us[t] - p["h"][t]
        J += dx.dot(p["Q"]).dot(dx) + du.dot(p["R"]).dot(du)
    dx = xs[p["T"]] - p["g"][p["T"]]
    return J + dx.dot(p["Qf"]).dot(dx)


def rollout(p, us):
    xs = [p["x0"]]
    for t in range(p["T"]):
        xs.append(p["F"].dot(xs[-1]) + p["G"].dot(us[t]))
    return np.array(xs)


def dense_optimum(p):
    """All controls at once: x_t is affine in U = (u_0 ... u_{T-1}); minimise the quadratic by solving
    its normal equations."""
    n, m, T, F, G = p["n"], p["m"], p["T"], p["F"], p["G"]
    Sx = np.zeros(((T + 1) * n, n))
    Su = np.zeros(((T + 1) * n, T * m))
    Fp = np.eye(n)
    for t in range(T + 1):
        Sx[t * n:(t + 1) * n] = Fp
        Fp = F.dot(Fp)
    for t in range(1, T + 1):
        for s in range(t):
            Su[t * n:(t + 1) * n, s * m:(s + 1) * m] = np.linalg.matrix_power(F, t - 1 - s).dot(G)
    Qs, Rs, Qfs = 0.5 * (p["Q"] + p["Q"].T), 0.5 * (p["R"] + p["R"].T), 0.5 * (p["Qf"] + p["Qf"].T)
    Qbig = np.zeros(((T + 1) * n, (T + 1) * n))
    for t in range(T):
        Qbig[t * n:(t + 1) * n, t * n:(t + 1) * n] = Qs
    Qbig[T * n:, T * n:] = Qfs
    Rbig = np.kron(np.eye(T), Rs)
    c = Sx.dot(p["x0"]) - p["g"].reshape(-1)
    H = Su.T.dot(Qbig).dot(Su) + Rbig
    rhs = -(Su.T.dot(Qbig).dot(c)) + Rbig.dot(p["h"].reshape(-1))
    U = np.linalg.solve(H, rhs).reshape(T, m)
    xs = rollout(p, U)
    return U, xs, cost(p, xs, U)


def riccati_gains(p):
    """K_t of u_t = K_t x_t + ...: K_t = -(R + G' P_{t+1} G)^-1 G' P_{t+1} F, P_t = Q + F' P_{t+1} (F + G K_t)."""
    Qs, Rs, Qfs = 0.5 * (p["Q"] + p["Q"].T), 0.5 * (p["R"] + p["R"].T), 0.5 * (p["Qf"] + p["Qf"].T)
    F, G = p["F"], p["G"]
    P = Qfs
    K = np.zeros((p["T"], p["m"], p["n"]))
    for t in range(p["T"] - 1, -1, -1):
        K[t] = -np.linalg.solve(Rs + G.T.dot(P).dot(G), G.T.dot(P).dot(F))
        P = Qs + F.T.dot(P).dot(F + G.dot(K[t]))
        P = 0.5 * (P + P.T)
    return K


def check(p, k, K, dJ1, dJ2, us_new, J_new, rtol=1e-10):
    """k [T,m], K [T,m,n], Delta_J1/2, and the result of the alpha = 1 forward pass, against numpy."""
    U, xs_opt, J_opt = dense_optimum(p)
    K_ric = riccati_gains(p)
    xs0 = rollout(p, p["us_start"])
    J0 = cost(p, xs0, p["us_start"])
    scale_K = np.abs(K_ric).max()
    assert np.abs(K - K_ric).max() <= rtol * scale_K * 10, np.abs(K - K_ric).max() / scale_K
    assert np.abs(us_new - U).max() <= 1e-9 * max(1.0, np.abs(U).max()), np.abs(us_new - U).max()
    assert abs(J_new - J_opt) <= rtol * abs(J_opt), (J_new, J_opt)
    # the expected improvement of the full step is the true one (exact model): -(dJ1 + dJ2) = J0 - J*
    assert abs(-(dJ1 + dJ2) - (J0 - J_opt)) <= 1e-9 * abs(J0 - J_opt), (-(dJ1 + dJ2), J0 - J_opt)
    # feed-forward term: along the starting trajectory u* - u = k + K (x* - x)
    return dict(J0=J0, J_opt=J_opt)
