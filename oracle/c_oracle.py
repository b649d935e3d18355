"""ctypes front end of the CPU oracle (``oracle/ilqr_oracle.c``).

TEST INFRASTRUCTURE: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` import this module.  The shipped solver
(``ilqr_amd``) never does.

All arrays are C-contiguous fp64 in the reference's own layout
(``xs [B, T+1, n]``, ``us [B, T, m]``, ``k [B, T, m]``, ``K [B, T, m, n]``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MODEL_CARTPOLE, MODEL_LINEAR, MODEL_SIGMOID, MODEL_USER = 0, 1, 2, 3
COST_QR, COST_PATHQR = 0, 1
MATH_LIBM, MATH_DET = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class _ModelC(C.Structure):
    _fields_ = [
        ("model", C.c_int), ("n", C.c_int), ("m", C.c_int), ("T", C.c_int),
        ("cost_kind", C.c_int), ("math_mode", C.c_int), ("control_limit", C.c_int), ("pad_", C.c_int),
        ("dt", C.c_double), ("mc", C.c_double), ("mp", C.c_double), ("l", C.c_double), ("g", C.c_double),
        ("A", _dp), ("B", _dp), ("Q", _dp), ("R", _dp), ("Qt", _dp),
        ("x_goal", _dp), ("u_goal", _dp), ("lower", _dp), ("upper", _dp),
        ("W", _dp), ("sig_c", C.c_double),
        ("user_f", C.c_void_p), ("user_jac", C.c_void_p), ("user_p", _dp),
    ]


class _OptsC(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int), ("tol", C.c_double), ("n_alpha", C.c_int), ("alphas", _dp),
        ("scale", C.c_double), ("mu_min", C.c_double), ("mu_max", C.c_double), ("delta0", C.c_double),
        ("qp_max_iter", C.c_int), ("qp_min_grad", C.c_double), ("qp_min_rel_improve", C.c_double),
        ("qp_step_dec", C.c_double), ("qp_min_step", C.c_double), ("qp_armijo", C.c_double),
    ]


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc, no FMA)."""
    src = [os.path.join(_HERE, f) for f in ("ilqr_oracle.c", "det_math.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def line_search_alphas(steps=30):
    """The reference's schedule, ilqr_core.py:267."""
    return 1.1 ** (-np.arange(steps) ** 2)


class Model:
    """Model + cost + bounds of one problem family (shared by a batch)."""

    def __init__(self, model=MODEL_CARTPOLE, n=5, m=1, T=100, cost_kind=COST_QR, math_mode=MATH_DET,
                 control_limit=True, dt=0.05, mc=1.0, mp=0.1, l=1.0, g=9.8, A=None, B=None,
                 Q=None, R=None, Q_terminal=None, x_goal=None, u_goal=None, lower=None, upper=None,
                 W=None, sig_c=0.0, user=None, user_params=None):
        self.n, self.m, self.T = n, m, T
        self.model, self.cost_kind, self.math_mode = model, cost_kind, math_mode
        self.control_limit = bool(control_limit)
        self.A = None if A is None else _f64(A).reshape(n, n)
        self.B = None if B is None else _f64(B).reshape(n, m)
        self.W = None if W is None else _f64(W).reshape(n, n)
        self.sig_c = float(sig_c)
        self.Q = _f64(Q).reshape(n, n)
        self.R = _f64(R).reshape(m, m)
        self.Qt = self.Q if Q_terminal is None else _f64(Q_terminal).reshape(n, n)
        if cost_kind == COST_QR:
            self.x_goal = _f64(np.zeros(n) if x_goal is None else x_goal).reshape(n)
            self.u_goal = _f64(np.zeros(m) if u_goal is None else u_goal).reshape(m)
        else:
            self.x_goal = _f64(x_goal).reshape(T + 1, n)
            self.u_goal = _f64(u_goal).reshape(T, m)
        self.lower = _f64(np.full(m, -np.inf) if lower is None else lower).reshape(m)
        self.upper = _f64(np.full(m, np.inf) if upper is None else upper).reshape(m)
        assert lib().oracle_dims_ok(n, m)
        c = _ModelC()
        c.model, c.n, c.m, c.T = model, n, m, T
        c.cost_kind, c.math_mode, c.control_limit = cost_kind, math_mode, int(self.control_limit)
        c.dt, c.mc, c.mp, c.l, c.g = dt, mc, mp, l, g
        c.A, c.B = _p(self.A), _p(self.B)
        c.W, c.sig_c = _p(self.W), self.sig_c
        if user is not None:  # (ctypes library of a user model, see build_user_model)
            self.user, self.user_params = user, _f64(user_params)
            c.user_f = C.cast(user.user_model_f, C.c_void_p)
            c.user_jac = C.cast(user.user_model_jac, C.c_void_p)
            c.user_p = _p(self.user_params)
        c.Q, c.R, c.Qt = _p(self.Q), _p(self.R), _p(self.Qt)
        c.x_goal, c.u_goal = _p(self.x_goal), _p(self.u_goal)
        c.lower, c.upper = _p(self.lower), _p(self.upper)
        self.c = c
        self.params = dict(dt=dt, mc=mc, mp=mp, l=l, g=g)
        self.dt = dt

    def with_math(self, math_mode):
        import copy
        other = copy.copy(self)
        c = _ModelC()
        C.memmove(C.byref(c), C.byref(self.c), C.sizeof(_ModelC))
        c.math_mode = math_mode
        other.c, other.math_mode = c, math_mode
        return other


class Opts:
    """update_u keyword arguments (ilqr_core.py:241) + the solver constants
    (:58-62) + the boxQP options the solver passes (:154-155)."""

    def __init__(self, max_iter=200, tol=1e-4, line_search_steps=30, scale=0.1, mu_min=1e-6,
                 mu_max=1e10, delta0=2.0, qp_max_iter=100, qp_min_grad=1e-8, qp_min_rel_improve=1e-8,
                 qp_step_dec=0.9, qp_min_step=1e-22, qp_armijo=0.1, alphas=None):
        self.alphas = _f64(line_search_alphas(line_search_steps) if alphas is None else alphas)
        o = _OptsC()
        o.max_iter, o.tol, o.n_alpha, o.alphas = max_iter, tol, len(self.alphas), _p(self.alphas)
        o.scale, o.mu_min, o.mu_max, o.delta0 = scale, mu_min, mu_max, delta0
        o.qp_max_iter, o.qp_min_grad, o.qp_min_rel_improve = qp_max_iter, qp_min_grad, qp_min_rel_improve
        o.qp_step_dec, o.qp_min_step, o.qp_armijo = qp_step_dec, qp_min_step, qp_armijo
        self.c = o
        self.max_iter = max_iter


def detmath(a, b):
    a, b = _f64(a).ravel(), _f64(b).ravel()
    s, c, t = np.empty_like(a), np.empty_like(a), np.empty_like(a)
    lib().oracle_detmath(len(a), _p(a), _p(b), _p(s), _p(c), _p(t))
    return s, c, t


def dynamics(M, x, u):
    x, u = _f64(x).reshape(-1, M.n), _f64(u).reshape(-1, M.m)
    cnt = x.shape[0]
    f, fx, fu = np.empty((cnt, M.n)), np.empty((cnt, M.n, M.n)), np.empty((cnt, M.n, M.m))
    lib().oracle_dynamics(C.byref(M.c), cnt, _p(x), _p(u), _p(f), _p(fx), _p(fu))
    return f, fx, fu


def derivatives(M, xs, us):
    """Per-step derivative tensors of ONE trajectory (xs [T+1,n], us [T,m])."""
    xs, us = _f64(xs).reshape(M.T + 1, M.n), _f64(us).reshape(M.T, M.m)
    T, n, m = M.T, M.n, M.m
    out = dict(fx=np.empty((T, n, n)), fu=np.empty((T, n, m)), lx=np.empty((T, n)), lu=np.empty((T, m)),
               lxx=np.empty((T, n, n)), luu=np.empty((T, m, m)), lux=np.empty((T, m, n)),
               lxT=np.empty(n), lxxT=np.empty((n, n)))
    lib().oracle_derivatives(C.byref(M.c), _p(xs), _p(us), *[_p(out[k]) for k in
                             ("fx", "fu", "lx", "lu", "lxx", "luu", "lux", "lxT", "lxxT")])
    return out


def rollout_cost(M, x0, us):
    x0, us = _f64(x0).reshape(-1, M.n), _f64(us).reshape(-1, M.T, M.m)
    B = x0.shape[0]
    xs, J = np.empty((B, M.T + 1, M.n)), np.empty(B)
    lib().oracle_rollout_cost(C.byref(M.c), B, _p(x0), _p(us), _p(xs), _p(J))
    return xs, J


def trajectory_cost(M, xs, us):
    xs, us = _f64(xs).reshape(-1, M.T + 1, M.n), _f64(us).reshape(-1, M.T, M.m)
    J = np.empty(xs.shape[0])
    lib().oracle_trajectory_cost(C.byref(M.c), xs.shape[0], _p(xs), _p(us), _p(J))
    return J


def backward(M, O, xs, us, mu):
    xs, us = _f64(xs).reshape(-1, M.T + 1, M.n), _f64(us).reshape(-1, M.T, M.m)
    B = xs.shape[0]
    mu = _f64(np.broadcast_to(mu, (B,)))
    k, K = np.empty((B, M.T, M.m)), np.empty((B, M.T, M.m, M.n))
    info = np.empty((B, 8))
    qp_result = np.empty((B, M.T), dtype=np.int32)
    free = np.empty((B, M.T, M.m), dtype=np.int32)
    lib().oracle_backward(C.byref(M.c), C.byref(O.c), B, _p(xs), _p(us), _p(mu), _p(k), _p(K), _p(info),
                          _pi(qp_result), _pi(free))
    return dict(k=k, K=K, dJ1=info[:, 0].copy(), dJ2=info[:, 1].copy(),
                stopped_at=info[:, 2].astype(np.int32), qp_iter=info[:, 3].astype(np.int32),
                qp_factor=info[:, 4].astype(np.int32), qp_backtrack=info[:, 5].astype(np.int32),
                n_clamped=info[:, 6].astype(np.int32), rc=info[:, 7].astype(np.int32),
                qp_result=qp_result, free=free)


def forward(M, alphas, xs, us, k, K):
    xs, us = _f64(xs).reshape(-1, M.T + 1, M.n), _f64(us).reshape(-1, M.T, M.m)
    B = xs.shape[0]
    k, K = _f64(k).reshape(B, M.T, M.m), _f64(K).reshape(B, M.T, M.m, M.n)
    alphas = _f64(alphas)
    nA = len(alphas)
    xs_new, us_new = np.empty((B, nA, M.T + 1, M.n)), np.empty((B, nA, M.T, M.m))
    J = np.empty((B, nA))
    lib().oracle_forward(C.byref(M.c), B, nA, _p(alphas), _p(xs), _p(us), _p(k), _p(K), _p(xs_new),
                         _p(us_new), _p(J))
    return xs_new, us_new, J


def boxqp(H, g, lower, upper, x0=None, O=None):
    """Batched standalone boxQP: H [B,m,m], g/lower/upper/x0 [B,m]."""
    O = O or Opts(qp_step_dec=0.6)
    H = _f64(H)
    m = H.shape[-1]
    H = H.reshape(-1, m, m)
    B = H.shape[0]
    g, lower, upper = (_f64(np.broadcast_to(a, (B, m))) for a in (g, lower, upper))
    x0 = None if x0 is None else _f64(np.broadcast_to(x0, (B, m)))
    x, Hff = np.empty((B, m)), np.empty((B, m, m))
    result, hd, rc = (np.empty(B, dtype=np.int32) for _ in range(3))
    free, cnt = np.empty((B, m), dtype=np.int32), np.empty((B, 3), dtype=np.int32)
    lib().oracle_boxqp(B, m, _p(H), _p(g), _p(lower), _p(upper), _p(x0), C.byref(O.c), _p(x), _pi(result),
                       _p(Hff), _pi(hd), _pi(free), _pi(cnt), _pi(rc))
    return dict(x=x, result=result, Hff=Hff, hff_dim=hd, free=free.astype(bool), n_iter=cnt[:, 0],
                n_factor=cnt[:, 1], n_backtrack=cnt[:, 2], rc=rc)


def solve(M, O, x0, us_init, delta=None, k=None, K=None, nthreads=1, trace=True):
    """Batched update_u.  Returns a dict; ``delta``/``k``/``K`` carry the solver
    state that persists across calls (defaults: a freshly constructed solver)."""
    x0, us_init = _f64(x0).reshape(-1, M.n), _f64(us_init).reshape(-1, M.T, M.m)
    B = x0.shape[0]
    delta = np.full(B, 2.0) if delta is None else _f64(delta).copy()
    k = np.zeros((B, M.T, M.m)) if k is None else _f64(k).copy()
    K = np.zeros((B, M.T, M.m, M.n)) if K is None else _f64(K).copy()
    xs, us = np.empty((B, M.T + 1, M.n)), np.empty((B, M.T, M.m))
    res_d, res_i = np.empty((B, 4)), np.empty((B, 4), dtype=np.int32)
    mi = O.max_iter
    tr = {}
    if trace:
        tr = dict(J=np.full((B, mi), np.nan), accepted=np.full((B, mi), -1, np.int32),
                  alpha_idx=np.full((B, mi), -1, np.int32), mu=np.full((B, mi), np.nan),
                  delta=np.full((B, mi), np.nan), ratio=np.full((B, mi), np.nan),
                  qp_iter=np.full((B, mi), -1, np.int32), qp_backtrack=np.full((B, mi), -1, np.int32),
                  n_clamped=np.full((B, mi), -1, np.int32), stopped_at=np.full((B, mi), -2, np.int32))
        lqr = np.empty((B, M.T), dtype=np.int32)
        lfree = np.empty((B, M.T, M.m), dtype=np.int32)
    else:
        lqr = lfree = None
    g = tr.get
    lib().oracle_solve(C.byref(M.c), C.byref(O.c), B, int(nthreads), _p(x0), _p(us_init), _p(delta), _p(k),
                       _p(K), _p(xs), _p(us), _p(res_d), _pi(res_i), _p(g("J")), _pi(g("accepted")),
                       _pi(g("alpha_idx")), _p(g("mu")), _p(g("delta")), _p(g("ratio")), _pi(g("qp_iter")),
                       _pi(g("qp_backtrack")), _pi(g("n_clamped")), _pi(g("stopped_at")), _pi(lqr),
                       _pi(lfree))
    out = dict(xs=xs, us=us, k=k, K=K, J=res_d[:, 0].copy(), mu=res_d[:, 1].copy(), delta=delta,
               iters=res_i[:, 0].copy(), converged=res_i[:, 1].copy(), status=res_i[:, 2].copy())
    if trace:
        out["trace"] = tr
        out["last_qp_result"], out["last_free"] = lqr, lfree
    return out


def solve_with_step_history(M, O, x0, us_init, **kw):
    """``solve`` of ONE problem that also returns, per iteration, what boxQP handed back at every time step of
    that iteration's backward pass: out["hist_qp_result"] [iters, T] (-100 = step not reached) and
    out["hist_free"] [iters, T, m]."""
    mi = O.max_iter
    hq = np.full((mi, M.T), -100, dtype=np.int32)
    hf = np.zeros((mi, M.T, M.m), dtype=np.int32)
    L = lib()
    L.oracle_set_step_history(_pi(hq), _pi(hf), C.c_long(mi))
    try:
        out = solve(M, O, np.asarray(x0).reshape(1, M.n), np.asarray(us_init).reshape(1, M.T, M.m), nthreads=1,
                    trace=True, **kw)
    finally:
        L.oracle_set_step_history(None, None, C.c_long(0))
    it = int(out["iters"][0])
    out["hist_qp_result"], out["hist_free"] = hq[:it], hf[:it].astype(bool)
    return out


def plant_step(st, u, gravity=9.8, masscart=1.0, masspole=0.1, length=0.5, tau=0.05, math_mode=MATH_DET):
    st, u = _f64(st).reshape(-1, 4), _f64(u).reshape(-1)
    par = _f64([gravity, masscart, masspole, length, tau])
    out = np.empty_like(st)
    lib().oracle_plant_step(st.shape[0], math_mode, _p(par), _p(st), _p(u), _p(out))
    return out


# ---- the workloads named by the benchmark configuration -------------------

def cartpole_model(T=100, math_mode=MATH_DET, dt=0.05, max_control=5.0, l=1.0, mpc=False):
    """examples/cartpole_example.py:11-32 (mpc=False) or
    examples/cartpole_example_mpc.py:84-100 (mpc=True)."""
    Q = 100.0 * np.eye(5)
    if mpc:
        Q[0, 0] = Q[1, 1] = 10.0
        Q[4, 4] = 80.0
        R = np.array([[1.0]])
    else:
        Q[1, 1] = Q[4, 4] = 10.0
        R = np.array([[5.0]])
    return Model(MODEL_CARTPOLE, 5, 1, T, COST_QR, math_mode, True, dt=dt, l=l, Q=Q, R=R,
                 Q_terminal=100.0 * np.eye(5), x_goal=[0.0, 0.0, 0.0, 1.0, 0.0],
                 lower=[-max_control], upper=[max_control])


def augment_state(st):
    st = np.asarray(st, dtype=np.float64)
    return np.stack([st[..., 0], st[..., 1], np.sin(st[..., 2]), np.cos(st[..., 2]), st[..., 3]], axis=-1)


def cartpole_batch(B, T=100, seed_x=1, seed_u=2):
    """Random swing-up batch: x, xd, thd ~ U(-1,1), theta ~ U(-pi,pi); us ~ U(-1,1)."""
    r = np.random.RandomState(seed_x)
    st = np.stack([r.uniform(-1, 1, B), r.uniform(-1, 1, B), r.uniform(-np.pi, np.pi, B),
                   r.uniform(-1, 1, B)], axis=1)
    us = np.random.RandomState(seed_u).uniform(-1, 1, (B, T, 1))
    return augment_state(st), us


def synthetic_model(n=32, m=8, T=200, seed=4, dt=0.05, sig_c=2.0, u_max=1.0, math_mode=MATH_DET):
    """The large-state workload (no counterpart in the reference, which ships no model of this
    size): x+ = x + dt (A x + B u + c phi(W x)), phi(z) = z / sqrt(1 + z^2), with a stable random
    A, random B and W, QRCost towards the origin and box limits |u| <= u_max."""
    r = np.random.RandomState(seed)
    G = r.randn(n, n)
    A = 0.8 * G / np.abs(np.linalg.eigvals(G)).max() - np.eye(n)
    Bm = r.randn(n, m) / np.sqrt(m)
    W = r.randn(n, n) / np.sqrt(n)
    Q, R, Qt = np.eye(n), 0.1 * np.eye(m), 10.0 * np.eye(n)
    return Model(MODEL_SIGMOID, n, m, T, COST_QR, math_mode, True, dt=dt, A=A, B=Bm, W=W, sig_c=sig_c, Q=Q, R=R,
                 Q_terminal=Qt, lower=np.full(m, -u_max), upper=np.full(m, u_max))


def synthetic_batch(B, n=32, m=8, T=200, seed_x=5, seed_u=6):
    x0 = np.random.RandomState(seed_x).uniform(-1.0, 1.0, (B, n))
    us = np.random.RandomState(seed_u).uniform(-0.5, 0.5, (B, T, m))
    return x0, us


def build_user_model(header, name):
    """Compile a user model header for the CPU (oracle/user_model_shim.c) and load it."""
    header = os.path.abspath(header)
    out = os.path.join(_HERE, "libusermodel_%s.so" % name)
    shim = os.path.join(_HERE, "user_model_shim.c")
    if not (os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(f) for f in (header, shim, os.path.join(_HERE, "..", "include", "ilqr_user_fd.h")))):
        subprocess.run(["gcc", "-O2", "-mfma", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-shared",
                        "-DILQR_USER_MODEL=\"%s\"" % header, "-I", _HERE, shim, "-o", out, "-lm"], check=True)
    return C.CDLL(out)


def user_model(lib_user, params, T, Q, R, Q_terminal=None, x_goal=None, u_goal=None, lower=None, upper=None,
               cost_kind=COST_QR):
    """cost_kind=COST_PATHQR: x_goal [T+1, n], u_goal [T, m] are the paths of a PathQRCost."""
    n, m = lib_user.user_model_n(), lib_user.user_model_m()
    assert lib_user.user_model_np() == len(params)
    return Model(MODEL_USER, n, m, T, cost_kind, MATH_DET, True, Q=Q, R=R, Q_terminal=Q_terminal, x_goal=x_goal,
                 u_goal=u_goal, lower=lower, upper=upper, user=lib_user, user_params=params)
