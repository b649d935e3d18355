"""ctypes binding of libilqr_hip.so (the C ABI declared in include/ilqr_hip.h).

PyTorch is used for device memory and streams only: every function here takes
contiguous fp64 / int32 ROCm tensors and passes their data_ptr() through the C
ABI.  There is no CPU fallback: without the shared library or without a GPU
these calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libilqr_hip.so")

ILQR_OK = 0
MODEL_CARTPOLE = 0
COST_QR, COST_PATH_QR = 0, 1
STATUS_RAN_OUT, STATUS_CONVERGED, STATUS_MU_MAX = 0, 1, 2

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p


class ModelC(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("T", C.c_int32),
        ("cost_kind", C.c_int32), ("control_limit", C.c_int32),
        ("dt", C.c_double), ("mc", C.c_double), ("mp", C.c_double), ("l", C.c_double), ("g", C.c_double),
        ("Q", _dp), ("R", _dp), ("Q_terminal", _dp), ("x_goal", _dp), ("u_goal", _dp),
        ("lower", _dp), ("upper", _dp),
        ("goal_stride_x", C.c_int64), ("goal_stride_u", C.c_int64),
    ]


class LargeModelC(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("T", C.c_int32), ("cost_kind", C.c_int32),
        ("dt", C.c_double), ("sig_c", C.c_double),
        ("A", _vp), ("B", _vp), ("W", _vp), ("Q", _vp), ("R", _vp), ("Q_terminal", _vp),
        ("x_goal", _vp), ("u_goal", _vp), ("lower", _vp), ("upper", _vp),
        ("goal_stride_x", C.c_int64), ("goal_stride_u", C.c_int64),
    ]


class OptsC(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int32), ("n_alpha", C.c_int32), ("alphas", _dp),
        ("tol", C.c_double), ("scale", C.c_double),
        ("mu_min", C.c_double), ("mu_max", C.c_double), ("delta0", C.c_double),
        ("qp_max_iter", C.c_int32), ("pad_", C.c_int32),
        ("qp_min_grad", C.c_double), ("qp_min_rel_improve", C.c_double), ("qp_step_dec", C.c_double),
        ("qp_min_step", C.c_double), ("qp_armijo", C.c_double), ("mu_init", C.c_double),
        ("dJ_carry", _vp),
    ]


class TraceC(C.Structure):
    _fields_ = [(name, _vp) for name in
                ("J", "accepted", "alpha_idx", "mu", "delta", "qp_iter", "qp_backtrack", "n_clamped",
                 "stopped_at")]


class DerivC(C.Structure):
    _fields_ = [(name, _vp) for name in ("fx", "fu", "lx", "lu", "lxx", "luu", "lux", "lxT", "lxxT")]


EXPORTS = (
    "ilqr_version", "ilqr_status_string", "ilqr_bm_ld", "ilqr_pack_bm", "ilqr_unpack_bm",
    "ilqr_rollout_cost_bm", "ilqr_deriv_bm", "ilqr_backward_bm", "ilqr_forward_ls_bm",
    "ilqr_boxqp_batched", "ilqr_solve_workspace_bytes", "ilqr_solve_batched",
    "ilqr_solve_chunked_workspace_bytes", "ilqr_solve_batched_chunked", "ilqr_solve_batched_static",
    "ilqr_cartpole_plant_step", "ilqr_detmath", "ilqr_debug_wave_log",
    "ilqr_large_rollout_cost", "ilqr_large_deriv", "ilqr_large_backward", "ilqr_large_forward_ls", "ilqr_large_workspace_bytes",
    "ilqr_large_solve_batched", "ilqr_large_debug_profile", "ilqr_cartpole_mpc_advance",
    "ilqr_xl_rollout_cost", "ilqr_xl_backward", "ilqr_xl_forward_ls", "ilqr_xl_workspace_bytes", "ilqr_xl_solve_batched",
)

_lib = None


class IlqrHipError(RuntimeError):
    pass


def lib():
    """Load libilqr_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IlqrHipError("%s is missing: build it with `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (hipcc, gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.ilqr_status_string.restype = C.c_char_p
        L.ilqr_bm_ld.restype = C.c_int64
        L.ilqr_bm_ld.argtypes = [C.c_int64]
        L.ilqr_solve_workspace_bytes.restype = C.c_size_t
        L.ilqr_solve_workspace_bytes.argtypes = [C.POINTER(ModelC), C.c_int64]
        L.ilqr_large_workspace_bytes.restype = C.c_size_t
        L.ilqr_large_workspace_bytes.argtypes = [C.POINTER(LargeModelC), C.c_int64]
        L.ilqr_xl_workspace_bytes.restype = C.c_size_t
        L.ilqr_xl_workspace_bytes.argtypes = [C.POINTER(LargeModelC), C.c_int64]
        _lib = L
    return _lib


def check(rc):
    if rc != ILQR_OK:
        raise IlqrHipError("libilqr_hip: %s (%d)" % (lib().ilqr_status_string(rc).decode(), rc))


def _host(a, shape):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))


def _ndim(a):
    return a.dim() if hasattr(a, "dim") else np.ndim(a)


def _dev64(a, device=None):
    """A contiguous fp64 tensor on the GPU holding ``a`` (numpy array or tensor on any device)."""
    import torch
    if not torch.is_tensor(a):
        a = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))
    if device is None:
        device = a.device if a.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return a.to(device=device, dtype=torch.float64).contiguous()


class Model:
    """Host description of the problem family (model + cost + bounds)."""

    def __init__(self, T, Q, R, Q_terminal=None, x_goal=None, u_goal=None, lower=None, upper=None,
                 dt=0.05, mc=1.0, mp=0.1, l=1.0, g=9.8, n=5, m=1, model=MODEL_CARTPOLE, control_limit=True,
                 x_path=None, u_path=None):
        """QRCost: x_goal [n], u_goal [m] (host).  PathQRCost: x_path [T+1, n], u_path [T, m] as
        fp64 device tensors (kept alive by this object); x_goal / u_goal are then ignored.
        One goal or one path PER PROBLEM: x_goal [B, n] and u_goal [B, m], or x_path [B, T+1, n] and
        u_path [B, T, m] (numpy or torch; copied to the GPU): problem b of every solve with this model
        tracks row b."""
        self.n, self.m, self.T = int(n), int(m), int(T)
        self.Q = _host(Q, (n, n))
        self.R = _host(R, (m, m))
        self.Qt = self.Q if Q_terminal is None else _host(Q_terminal, (n, n))
        self.batch_goals = None  # number of problems the per-problem goals are for
        per_problem_goal = x_path is None and x_goal is not None and _ndim(x_goal) == 2
        self.x_goal = _host(np.zeros(n) if (x_goal is None or per_problem_goal) else x_goal, (n,))
        self.u_goal = _host(np.zeros(m) if (u_goal is None or per_problem_goal) else u_goal, (m,))
        self.lower = _host(lower, (m,))
        self.upper = _host(upper, (m,))
        c = ModelC()
        c.model, c.n, c.m, c.T = model, n, m, T
        c.cost_kind, c.control_limit = COST_QR, int(bool(control_limit))
        c.dt, c.mc, c.mp, c.l, c.g = dt, mc, mp, l, g
        for name, arr in (("Q", self.Q), ("R", self.R), ("Q_terminal", self.Qt), ("x_goal", self.x_goal),
                          ("u_goal", self.u_goal), ("lower", self.lower), ("upper", self.upper)):
            setattr(c, name, arr.ctypes.data_as(_dp))
        if x_path is not None:
            import torch
            if u_path is None:
                raise ValueError("PathQRCost needs both x_path and u_path")
            if _ndim(x_path) == 3:  # one path per problem
                self.x_path, self.u_path = _dev64(x_path), _dev64(u_path)
                Bg = self.x_path.shape[0]
                if tuple(self.x_path.shape) != (Bg, T + 1, n) or tuple(self.u_path.shape) != (Bg, T, m):
                    raise ValueError("per-problem paths must be x_path [B, %d, %d] and u_path [B, %d, %d]"
                                     % (T + 1, n, T, m))
                self.batch_goals = int(Bg)
                c.goal_stride_x, c.goal_stride_u = (T + 1) * n, T * m
            else:
                self.x_path = _req(x_path, torch.float64)
                self.u_path = _req(u_path, torch.float64)
                if tuple(self.x_path.shape) != (T + 1, n) or tuple(self.u_path.shape) != (T, m):
                    raise ValueError("x_path must be [%d, %d] and u_path [%d, %d]" % (T + 1, n, T, m))
            c.cost_kind = COST_PATH_QR
            c.x_goal = C.cast(C.c_void_p(self.x_path.data_ptr()), _dp)
            c.u_goal = C.cast(C.c_void_p(self.u_path.data_ptr()), _dp)
        elif per_problem_goal:  # one QRCost goal per problem
            self.x_goals = _dev64(x_goal)
            Bg = self.x_goals.shape[0]
            self.u_goals = _dev64(np.zeros((Bg, m)) if u_goal is None else u_goal)
            if tuple(self.x_goals.shape) != (Bg, n) or tuple(self.u_goals.shape) != (Bg, m):
                raise ValueError("per-problem goals must be x_goal [B, %d] and u_goal [B, %d]" % (n, m))
            self.batch_goals = int(Bg)
            c.goal_stride_x, c.goal_stride_u = n, m
            c.x_goal = C.cast(C.c_void_p(self.x_goals.data_ptr()), _dp)
            c.u_goal = C.cast(C.c_void_p(self.u_goals.data_ptr()), _dp)
        self.c = c

    def subset(self, idx):
        """The model for the sub-batch ``idx`` (device index tensor) of the problems it was built for: the same
        object unless every problem has its own goal or path, else a copy with those gathered."""
        if self.batch_goals is None:
            return self
        import copy
        other = copy.copy(self)
        c = ModelC()
        C.memmove(C.byref(c), C.byref(self.c), C.sizeof(ModelC))
        if self.c.cost_kind == COST_PATH_QR:
            other.x_path, other.u_path = self.x_path[idx].contiguous(), self.u_path[idx].contiguous()
            gx, gu = other.x_path, other.u_path
        else:
            other.x_goals, other.u_goals = self.x_goals[idx].contiguous(), self.u_goals[idx].contiguous()
            gx, gu = other.x_goals, other.u_goals
        other.batch_goals = int(gx.shape[0])
        c.x_goal = C.cast(C.c_void_p(gx.data_ptr()), _dp)
        c.u_goal = C.cast(C.c_void_p(gu.data_ptr()), _dp)
        other.c = c
        return other


class Opts:
    def __init__(self, max_iter=200, tol=1e-4, line_search_steps=30, scale=0.1, mu_min=1e-6, mu_max=1e10,
                 delta0=2.0, qp_max_iter=100, qp_min_grad=1e-8, qp_min_rel_improve=1e-8, qp_step_dec=0.9,
                 qp_min_step=1e-22, qp_armijo=0.1, alphas=None, mu_from_array=False):
        # the reference's schedule, ilqr_core.py:267
        self.alphas = _host(1.1 ** (-np.arange(line_search_steps) ** 2) if alphas is None else alphas, (-1,))
        o = OptsC()
        o.max_iter, o.n_alpha = int(max_iter), len(self.alphas)
        o.alphas = self.alphas.ctypes.data_as(_dp)
        o.tol, o.scale = tol, scale
        o.mu_min, o.mu_max, o.delta0 = mu_min, mu_max, delta0
        o.qp_max_iter = qp_max_iter
        o.qp_min_grad, o.qp_min_rel_improve, o.qp_step_dec = qp_min_grad, qp_min_rel_improve, qp_step_dec
        o.qp_min_step, o.qp_armijo = qp_min_step, qp_armijo
        o.mu_init = 0.0 if mu_from_array else -1.0
        o.dJ_carry = None
        self.c = o
        self.max_iter = int(max_iter)

    def carrying(self, dJ):
        """A copy of the C struct for ONE call, with ``dJ_carry`` pointing at ``dJ`` ([2, B] fp64 device tensor or
        None): the expected-improvement sums Delta_J1 / Delta_J2 of a solve that is cut into several calls.  The
        shared struct is never modified (no pointer to a caller's temporary outlives the call), and the copy keeps
        the tensor and the step-size array alive for as long as it exists."""
        import torch
        if dJ is not None:
            assert dJ.dim() == 2 and dJ.shape[0] == 2
            _req(dJ, torch.float64)
        c = OptsC()
        C.memmove(C.byref(c), C.byref(self.c), C.sizeof(OptsC))
        c.dJ_carry = None if dJ is None else dJ.data_ptr()
        c.keep_alive_ = (self.alphas, dJ)
        return c


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, dtype):
    import torch
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, (t.device, t.dtype, t.is_contiguous())
    return t


def bm_ld(B):
    return int(lib().ilqr_bm_ld(int(B)))


def pack_bm(src, ld=None):
    """ref [B, ...] -> bm [R, ld] (R = product of the trailing dims)."""
    import torch
    B = src.shape[0]
    src = _req(src.reshape(B, -1).contiguous(), torch.float64)
    R = src.shape[1]
    ld = bm_ld(B) if ld is None else ld
    dst = torch.zeros((R, ld), dtype=torch.float64, device=src.device)
    check(lib().ilqr_pack_bm(_ptr(src), _ptr(dst), C.c_int64(B), C.c_int64(R), C.c_int64(ld), _stream()))
    return dst


def unpack_bm(src_bm, B, shape):
    import torch
    src_bm = _req(src_bm, torch.float64)
    R, ld = src_bm.shape
    dst = torch.empty((B, R), dtype=torch.float64, device=src_bm.device)
    check(lib().ilqr_unpack_bm(_ptr(src_bm), _ptr(dst), C.c_int64(B), C.c_int64(R), C.c_int64(ld), _stream()))
    return dst.reshape((B,) + tuple(shape))


def rollout_cost_bm(M, B, x0_bm, us_bm):
    import torch
    ld = x0_bm.shape[1]
    xs = torch.empty(((M.T + 1) * M.n, ld), dtype=torch.float64, device=x0_bm.device)
    J = torch.empty(B, dtype=torch.float64, device=x0_bm.device)
    check(lib().ilqr_rollout_cost_bm(C.byref(M.c), C.c_int64(B), C.c_int64(ld), _ptr(x0_bm), _ptr(us_bm),
                                     _ptr(xs), _ptr(J), _stream()))
    return xs, J


def deriv_bm(M, B, xs_bm, us_bm):
    import torch
    ld = xs_bm.shape[1]
    T, n, m = M.T, M.n, M.m
    rows = dict(fx=T * n * n, fu=T * n * m, lx=T * n, lu=T * m, lxx=T * n * n, luu=T * m * m, lux=T * m * n,
                lxT=n, lxxT=n * n)
    out = {k: torch.empty((r, ld), dtype=torch.float64, device=xs_bm.device) for k, r in rows.items()}
    d = DerivC()
    for k in rows:
        setattr(d, k, out[k].data_ptr())
    check(lib().ilqr_deriv_bm(C.byref(M.c), C.c_int64(B), C.c_int64(ld), _ptr(xs_bm), _ptr(us_bm), C.byref(d),
                              _stream()))
    return out


def backward_bm(M, O, B, xs_bm, us_bm, mu, want_masks=True, deriv=None):
    """deriv: the dict returned by deriv_bm -> the staged sweep that reads the derivative streams."""
    import torch
    ld = us_bm.shape[1]
    dev = us_bm.device
    d_c = None
    if deriv is not None:
        d_c = DerivC()
        for name in ("fx", "fu", "lx", "lu", "lxx", "luu", "lux", "lxT", "lxxT"):
            setattr(d_c, name, deriv[name].data_ptr())
    k = torch.empty((M.T * M.m, ld), dtype=torch.float64, device=dev)
    K = torch.empty((M.T * M.m * M.n, ld), dtype=torch.float64, device=dev)
    dJ = torch.zeros((2, ld), dtype=torch.float64, device=dev)
    info = torch.zeros((5, ld), dtype=torch.int32, device=dev)
    qres = torch.full((M.T, ld), -100, dtype=torch.int32, device=dev) if want_masks else None
    free = torch.zeros((M.T * M.m, ld), dtype=torch.int32, device=dev) if want_masks else None
    check(lib().ilqr_backward_bm(C.byref(M.c), C.byref(O.c), C.c_int64(B), C.c_int64(ld), _ptr(xs_bm),
                                 _ptr(us_bm), _ptr(_req(mu, torch.float64)),
                                 C.byref(d_c) if d_c is not None else None, _ptr(k), _ptr(K), _ptr(dJ),
                                 _ptr(info), _ptr(qres), _ptr(free), _stream()))
    return dict(k=k, K=K, dJ=dJ, info=info, qp_result=qres, free=free)


def forward_ls_bm(M, B, alphas, xs_bm, us_bm, k_bm, K_bm, want_traj=False):
    import torch
    ld = xs_bm.shape[1]
    dev = xs_bm.device
    al = _host(alphas, (-1,))
    nA = len(al)
    J = torch.empty((nA, ld), dtype=torch.float64, device=dev)
    xs_new = torch.empty((nA, (M.T + 1) * M.n, ld), dtype=torch.float64, device=dev) if want_traj else None
    us_new = torch.empty((nA, M.T * M.m, ld), dtype=torch.float64, device=dev) if want_traj else None
    check(lib().ilqr_forward_ls_bm(C.byref(M.c), C.c_int64(B), C.c_int64(ld), C.c_int32(nA),
                                   al.ctypes.data_as(_dp), _ptr(xs_bm), _ptr(us_bm), _ptr(k_bm), _ptr(K_bm),
                                   _ptr(J), _ptr(xs_new), _ptr(us_new), _stream()))
    return J, xs_new, us_new


def solve_workspace_bytes(M, B):
    return int(lib().ilqr_solve_workspace_bytes(C.byref(M.c), C.c_int64(B)))


TRACE_FIELDS = (("J", "f8"), ("accepted", "i4"), ("alpha_idx", "i4"), ("mu", "f8"), ("delta", "f8"),
                ("qp_iter", "i4"), ("qp_backtrack", "i4"), ("n_clamped", "i4"), ("stopped_at", "i4"))


def solve(M, O, x0, us_init, delta=None, k=None, K=None, trace=False, workspace=None, out=None, mu=None, dJ=None):
    """ilqr_solve_batched on ref-layout tensors.  Returns a dict of device tensors.

    ``out`` may carry preallocated result tensors (keys xs, us, J, mu, iters,
    status) and ``workspace`` a uint8 tensor, so that a timed loop allocates nothing.
    ``dJ`` [2, B]: Delta_J1 / Delta_J2 carried between the calls of a solve cut into several (in/out).
    """
    import torch
    B = x0.shape[0]
    _check_goal_batch(M, B)
    dev = x0.device
    T, n, m = M.T, M.n, M.m
    x0 = _req(x0.reshape(B, n), torch.float64)
    us_init = _req(us_init.reshape(B, T, m), torch.float64)
    delta = torch.full((B,), 2.0, dtype=torch.float64, device=dev) if delta is None else delta
    k = torch.zeros((B, T, m), dtype=torch.float64, device=dev) if k is None else k
    K = torch.zeros((B, T, m, n), dtype=torch.float64, device=dev) if K is None else K
    out = {} if out is None else out
    res = dict(
        xs=out.get("xs", None), us=out.get("us", None), J=out.get("J", None), mu=out.get("mu", None),
        iters=out.get("iters", None), status=out.get("status", None))
    if res["xs"] is None:
        res["xs"] = torch.empty((B, T + 1, n), dtype=torch.float64, device=dev)
        res["us"] = torch.empty((B, T, m), dtype=torch.float64, device=dev)
        res["J"] = torch.empty(B, dtype=torch.float64, device=dev)
        res["mu"] = torch.ones(B, dtype=torch.float64, device=dev) if mu is None else mu
        res["iters"] = torch.empty(B, dtype=torch.int32, device=dev)
        res["status"] = torch.empty(B, dtype=torch.int32, device=dev)
    if mu is not None:
        res["mu"] = mu
    need = solve_workspace_bytes(M, B)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    tr_c, tr = None, None
    if trace:
        tr, tr_c = {}, TraceC()
        for name, dt in TRACE_FIELDS:
            if dt == "f8":
                t = torch.full((O.max_iter, B), float("nan"), dtype=torch.float64, device=dev)
            else:
                t = torch.full((O.max_iter, B), -1, dtype=torch.int32, device=dev)
            tr[name] = t
            setattr(tr_c, name, t.data_ptr())
    check(lib().ilqr_solve_batched(
        C.byref(M.c), C.byref(O.carrying(dJ)), C.c_int64(B), _ptr(x0), _ptr(us_init), _ptr(_req(delta, torch.float64)),
        _ptr(_req(k, torch.float64)), _ptr(_req(K, torch.float64)), _ptr(res["xs"]), _ptr(res["us"]),
        _ptr(res["J"]), _ptr(res["mu"]), _ptr(res["iters"]), _ptr(res["status"]),
        C.byref(tr_c) if tr_c is not None else None, _ptr(workspace), C.c_size_t(workspace.numel()), _stream()))
    res.update(delta=delta, k=k, K=K, trace=tr, workspace=workspace)
    return res


STATIC_WORKSPACE_BYTES = 32 << 20


def solve_static(M, O, x0, us_init, delta, k, K, out, trace=None):
    """ilqr_solve_batched_static: the solve on the library's own fixed workspace; the call allocates nothing.
    Every tensor is the caller's: delta [B], k [B,T,m], K [B,T,m,n] (in/out) and ``out`` with xs, us, J, mu,
    iters, status.  ``trace``: None or a dict of [max_iter, B] tensors keyed as TRACE_FIELDS."""
    import torch
    B = x0.shape[0]
    _check_goal_batch(M, B)
    f64, i32 = torch.float64, torch.int32
    tr_c = None
    if trace is not None:
        tr_c = TraceC()
        for name, _ in TRACE_FIELDS:
            setattr(tr_c, name, trace[name].data_ptr())
    check(lib().ilqr_solve_batched_static(
        C.byref(M.c), C.byref(O.carrying(None)), C.c_int64(B), _ptr(_req(x0, f64)), _ptr(_req(us_init, f64)),
        _ptr(_req(delta, f64)), _ptr(_req(k, f64)), _ptr(_req(K, f64)), _ptr(_req(out["xs"], f64)),
        _ptr(_req(out["us"], f64)), _ptr(_req(out["J"], f64)), _ptr(_req(out["mu"], f64)),
        _ptr(_req(out["iters"], i32)), _ptr(_req(out["status"], i32)), C.byref(tr_c) if tr_c is not None else None,
        _stream()))
    return out


def solve_chunked(M, O, x0, us_init, chunk, delta=None, k=None, K=None, workspace=None, out=None):
    """ilqr_solve_batched_chunked: launches of ``chunk`` iterations, finished problems dropped in
    between.  Same dict as ``solve`` (no trace); ``out`` may carry preallocated result tensors
    (xs, us, J, mu, iters, status)."""
    import torch
    B, dev = x0.shape[0], x0.device
    _check_goal_batch(M, B)
    T, n, m = M.T, M.n, M.m
    f64 = torch.float64
    x0 = _req(x0.reshape(B, n), f64)
    us_init = _req(us_init.reshape(B, T, m), f64)
    delta = torch.full((B,), 2.0, dtype=f64, device=dev) if delta is None else delta
    k = torch.zeros((B, T, m), dtype=f64, device=dev) if k is None else k
    K = torch.zeros((B, T, m, n), dtype=f64, device=dev) if K is None else K
    if out is not None:
        res = {name: out[name] for name in ("xs", "us", "J", "mu", "iters", "status")}
        for name, t in res.items():
            _req(t, torch.int32 if name in ("iters", "status") else f64)
            assert t.shape[0] == B, name
    else:
        res = dict(xs=torch.empty((B, T + 1, n), dtype=f64, device=dev), us=torch.empty((B, T, m), dtype=f64, device=dev),
                   J=torch.empty(B, dtype=f64, device=dev), mu=torch.empty(B, dtype=f64, device=dev),
                   iters=torch.empty(B, dtype=torch.int32, device=dev),
                   status=torch.empty(B, dtype=torch.int32, device=dev))
    L = lib()
    L.ilqr_solve_chunked_workspace_bytes.restype = C.c_size_t
    need = int(L.ilqr_solve_chunked_workspace_bytes(C.byref(M.c), C.c_int64(B)))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    check(L.ilqr_solve_batched_chunked(
        C.byref(M.c), C.byref(O.carrying(None)), C.c_int32(int(chunk)), C.c_int64(B), _ptr(x0), _ptr(us_init),
        _ptr(_req(delta, f64)), _ptr(_req(k, f64)), _ptr(_req(K, f64)), _ptr(res["xs"]), _ptr(res["us"]), _ptr(res["J"]),
        _ptr(res["mu"]), _ptr(res["iters"]), _ptr(res["status"]), _ptr(workspace), C.c_size_t(workspace.numel()),
        _stream()))
    res.update(delta=delta, k=k, K=K, trace=None, workspace=workspace)
    return res


def boxqp_batched(H, g, lower, upper, x0, O):
    import torch
    B, m = g.shape
    dev = g.device
    x = torch.empty((B, m), dtype=torch.float64, device=dev)
    Hff = torch.empty((B, m, m), dtype=torch.float64, device=dev)
    result = torch.empty(B, dtype=torch.int32, device=dev)
    hd = torch.empty(B, dtype=torch.int32, device=dev)
    free = torch.empty((B, m), dtype=torch.int32, device=dev)
    cnt = torch.empty((B, 3), dtype=torch.int32, device=dev)
    check(lib().ilqr_boxqp_batched(C.c_int64(B), C.c_int32(m), _ptr(_req(H, torch.float64)),
                                   _ptr(_req(g, torch.float64)), _ptr(_req(lower, torch.float64)),
                                   _ptr(_req(upper, torch.float64)),
                                   _ptr(None if x0 is None else _req(x0, torch.float64)), C.byref(O.c), _ptr(x),
                                   _ptr(result), _ptr(Hff), _ptr(hd), _ptr(free), _ptr(cnt), _stream()))
    return dict(x=x, result=result, Hff=Hff, hff_dim=hd, free=free, counters=cnt)


def plant_step(st, u, gravity=9.8, masscart=1.0, masspole=0.1, length=0.5, tau=0.05):
    import torch
    B = st.shape[0]
    par = _host([gravity, masscart, masspole, length, tau], (5,))
    out = torch.empty_like(st)
    check(lib().ilqr_cartpole_plant_step(C.c_int64(B), par.ctypes.data_as(_dp), _ptr(_req(st, torch.float64)),
                                         _ptr(_req(u.reshape(B), torch.float64)), _ptr(out), _stream()))
    return out


def mpc_advance(st, us_solved, us_seed, u_applied, x0_aug, gravity=9.8, masscart=1.0, masspole=0.1, length=0.5,
                tau=0.05):
    """Plant step with us_solved[:, 0] (in place on st), augmented state and shifted control seed for the next
    solve, one launch.  All arguments are contiguous fp64 device tensors; us_seed must not be us_solved."""
    import torch
    B, N = us_solved.shape[0], us_solved.shape[1]
    for t in (st, us_solved, us_seed, u_applied, x0_aug):
        if t.dtype != torch.float64 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("mpc_advance: contiguous fp64 device tensors expected")
    if st.shape != (B, 4) or us_seed.numel() != B * N or u_applied.numel() != B or x0_aug.shape != (B, 5):
        raise ValueError("mpc_advance: shapes")
    par = _host([gravity, masscart, masspole, length, tau], (5,))
    check(lib().ilqr_cartpole_mpc_advance(C.c_int64(B), C.c_int64(N), par.ctypes.data_as(_dp), _ptr(st),
                                          _ptr(us_solved), _ptr(us_seed), _ptr(u_applied), _ptr(x0_aug), _stream()))


def detmath(a, b):
    import torch
    n = a.numel()
    outs = [torch.empty_like(a) for _ in range(5)]
    check(lib().ilqr_detmath(C.c_int64(n), _ptr(_req(a, torch.float64)), _ptr(_req(b, torch.float64)),
                             *[_ptr(o) for o in outs], _stream()))
    return outs


# ---- large-state path (n = 32, m = 8, one problem per wavefront) -----------------------------

def pad_last(t, size, dims=1):
    """``t`` with its last ``dims`` dimensions zero-padded to ``size`` (an int, or one int per dimension)."""
    import torch
    sizes = (size,) * dims if isinstance(size, int) else tuple(size)
    if tuple(t.shape[-dims:]) == sizes:
        return t.contiguous()
    out = torch.zeros(tuple(t.shape[:-dims]) + sizes, dtype=t.dtype, device=t.device)
    out[(Ellipsis,) + tuple(slice(0, s) for s in t.shape[-dims:])] = t
    return out


LARGE_N, LARGE_M = 32, 8  # the sizes the register-resident kernels work on (smaller models run zero-padded)
XL_N, XL_M = 64, 16       # the largest model of the LDS-resident kernels


class LargeModel:
    """Synthetic model x+ = x + dt (A x + B u + c phi(W x)) with a QRCost and box limits; every
    array is uploaded once and kept alive by this object."""

    def __init__(self, T, A, B, W, Q, R, Q_terminal=None, x_goal=None, u_goal=None, lower=None, upper=None,
                 dt=0.05, sig_c=2.0, device=None, x_path=None, u_path=None, user=None, wide=False):
        """A [n, n], B [n, m], W [n, n] with n <= 32 states and m <= 8 controls (the kernels work on 32 x 8; smaller
        models are padded with zeros here, which leaves every result bit the same, see ilqr_large_model).
        x_path [T+1, n], u_path [T, m]: the goals of a PathQRCost instead of x_goal / u_goal.
        One goal or path per problem: x_goal [B, n], u_goal [B, m] or x_path [B, T+1, n], u_path [B, T, m].
        user = (library, params, n, m): the dynamics are those of a per-model library built by
        ``user_model.build_large_user_model`` (A, B, W are then ignored).
        Models with more than 32 states or 8 controls (up to 64 and 16) run on the LDS-resident kernels
        (ilqr_xl_*), dense arrays; ``wide=True`` puts a smaller model there as well (same results)."""
        import torch
        if user is not None:
            self.lib, n_real, m_real = user[0], int(user[2]), int(user[3])
        else:
            n_real, m_real = int(np.shape(A)[0]), int(np.shape(B)[1])
        if not (1 <= n_real <= XL_N and 1 <= m_real <= XL_M):
            raise ValueError("the large-state paths hold up to %d states and %d controls, not %d and %d"
                             % (XL_N, XL_M, n_real, m_real))
        # up to 32 x 8: the register-resident kernels, which work on 32 x 8 (smaller models zero-padded);
        # beyond: the LDS-resident kernels (ilqr_xl_*), dense arrays of the model's own size
        self.xl = bool(wide) or n_real > LARGE_N or m_real > LARGE_M
        if self.xl and user is not None:
            raise ValueError("per-model libraries hold up to 32 states and 8 controls")
        n, m = (n_real, m_real) if self.xl else (LARGE_N, LARGE_M)
        self.full = (n, m)
        self.n, self.m, self.T = n_real, m_real, int(T)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device

        nr, mr, T1 = n_real, m_real, int(T) + 1

        def up(a, real, full):
            """the array (of shape ``real``), zero-padded to ``full``, on the device"""
            out = np.zeros(full)
            out[tuple(slice(0, r) for r in real)] = _host(a, real)
            return torch.tensor(out, device=dev)
        self.batch_goals = None
        per_problem_goal = x_path is None and x_goal is not None and _ndim(x_goal) == 2
        if user is not None:  # the model's parameter vector travels in A; B and W are not read
            p = _host(user[1], (-1,))
            A, B, W = None, None, None
        self.t = dict(
            A=torch.tensor(p, device=dev) if user is not None else up(A, (nr, nr), (n, n)),
            B=None if user is not None else up(B, (nr, mr), (n, m)),
            W=None if user is not None else up(W, (nr, nr), (n, n)), Q=up(Q, (nr, nr), (n, n)),
            R=up(R, (mr, mr), (m, m)), Q_terminal=up(Q if Q_terminal is None else Q_terminal, (nr, nr), (n, n)),
            x_goal=up(np.zeros(nr) if (x_goal is None or per_problem_goal) else x_goal, (nr,), (n,)),
            u_goal=up(np.zeros(mr) if (u_goal is None or per_problem_goal) else u_goal, (mr,), (m,)),
            lower=up(lower, (mr,), (m,)), upper=up(upper, (mr,), (m,)))
        c = LargeModelC()
        c.n, c.m, c.T, c.dt, c.sig_c = n_real, m_real, int(T), dt, sig_c
        c.cost_kind = COST_QR
        if x_path is not None or u_path is not None:
            if x_path is None or u_path is None:
                raise ValueError("PathQRCost needs both x_path and u_path")
            if _ndim(x_path) == 3:  # one path per problem
                self.t["x_goal"], self.t["u_goal"] = pad_last(_dev64(x_path, dev), n), pad_last(_dev64(u_path, dev), m)
                Bg = self.t["x_goal"].shape[0]
                if tuple(self.t["x_goal"].shape) != (Bg, int(T) + 1, n) or tuple(self.t["u_goal"].shape) != (Bg, int(T), m):
                    raise ValueError("per-problem paths must be x_path [B, T+1, n] and u_path [B, T, m]")
                self.batch_goals = int(Bg)
                c.goal_stride_x, c.goal_stride_u = (int(T) + 1) * n, int(T) * m
            else:
                self.t["x_goal"] = up(x_path, (T1, nr), (T1, n))
                self.t["u_goal"] = up(u_path, (T1 - 1, mr), (T1 - 1, m))
            c.cost_kind = COST_PATH_QR
        elif per_problem_goal:
            self.t["x_goal"] = pad_last(_dev64(x_goal, dev), n)
            Bg = self.t["x_goal"].shape[0]
            self.t["u_goal"] = pad_last(_dev64(np.zeros((Bg, m)) if u_goal is None else u_goal, dev), m)
            if tuple(self.t["x_goal"].shape) != (Bg, n) or tuple(self.t["u_goal"].shape) != (Bg, m):
                raise ValueError("per-problem goals must be x_goal [B, n] and u_goal [B, m]")
            self.batch_goals = int(Bg)
            c.goal_stride_x, c.goal_stride_u = n, m
        for name, t in self.t.items():
            setattr(c, name, None if t is None else t.data_ptr())
        self.c = c

    def subset(self, idx):
        """The model for the sub-batch ``idx`` (device index tensor) of the problems it was built for: the same
        object unless every problem has its own goal, else a copy with those goals gathered."""
        if self.batch_goals is None:
            return self
        import copy
        other = copy.copy(self)
        other.t = dict(self.t)
        other.t["x_goal"] = self.t["x_goal"][idx].contiguous()
        other.t["u_goal"] = self.t["u_goal"][idx].contiguous()
        other.batch_goals = int(other.t["x_goal"].shape[0])
        c = LargeModelC()
        C.memmove(C.byref(c), C.byref(self.c), C.sizeof(LargeModelC))
        c.x_goal, c.u_goal = other.t["x_goal"].data_ptr(), other.t["u_goal"].data_ptr()
        other.c = c
        return other


def _check_goal_batch(M, B):
    if getattr(M, "batch_goals", None) is not None and M.batch_goals != B:
        raise ValueError("the model carries goals for %d problems, the batch has %d (Model.subset(idx) gives the "
                         "model of a sub-batch)" % (M.batch_goals, B))


def _large_lib(M):
    """The library that holds the model's kernels: its own (a model compiled from a C source) or the main one."""
    own = getattr(M, "lib", None)
    if own is None:
        return lib()
    own.ilqr_large_workspace_bytes.restype = C.c_size_t
    return own


def _large_fn(M, name):
    """The entry point `name` of the family that holds the model: ilqr_large_* (up to 32 x 8, zero-padded layout) or
    ilqr_xl_* (up to 64 x 16, dense layout)."""
    return getattr(_large_lib(M), ("ilqr_xl_" if getattr(M, "xl", False) else "ilqr_large_") + name)


def _large_full(M):
    return getattr(M, "full", (LARGE_N, LARGE_M))


def large_rollout_cost(M, x0, us):
    import torch
    B = x0.shape[0]
    FN, FM = _large_full(M)
    x0, us = pad_last(x0, FN), pad_last(us, FM)
    xs = torch.empty((B, M.T + 1, FN), dtype=torch.float64, device=x0.device)
    J = torch.empty(B, dtype=torch.float64, device=x0.device)
    check(_large_fn(M, "rollout_cost")(C.byref(M.c), C.c_int64(B), _ptr(_req(x0, torch.float64)),
                                        _ptr(_req(us, torch.float64)), _ptr(xs), _ptr(J), _stream()))
    return (xs if M.n == FN else xs[..., :M.n].contiguous()), J


def large_deriv(M, xs, us, out=None):
    """ilqr_large_deriv: the derivative tensors of every (problem, step) in the reference's layout."""
    import torch
    B, dev = xs.shape[0], xs.device
    T, n, m = M.T, M.n, M.m
    shapes = dict(fx=(B, T, n, n), fu=(B, T, n, m), lx=(B, T, n), lu=(B, T, m), lxx=(B, T, n, n), luu=(B, T, m, m),
                  lux=(B, T, m, n), lxT=(B, n), lxxT=(B, n, n))
    if out is None:
        out = {k: torch.empty(s, dtype=torch.float64, device=dev) for k, s in shapes.items()}
    d = DerivC()
    for k, s in shapes.items():
        assert tuple(out[k].shape) == s
        setattr(d, k, _req(out[k], torch.float64).data_ptr())
    check(_large_lib(M).ilqr_large_deriv(C.byref(M.c), C.c_int64(B), _ptr(_req(xs, torch.float64)),
                                 _ptr(_req(us, torch.float64)), C.byref(d), _stream()))
    return out


def large_backward(M, O, xs, us, mu, want_masks=True):
    import torch
    B, dev = xs.shape[0], xs.device
    FN, FM = _large_full(M)
    xs, us = pad_last(xs, FN), pad_last(us, FM)
    k = torch.empty((B, M.T, FM), dtype=torch.float64, device=dev)
    K = torch.empty((B, M.T, FM, FN), dtype=torch.float64, device=dev)
    dJ = torch.empty((2, B), dtype=torch.float64, device=dev)
    info = torch.empty((5, B), dtype=torch.int32, device=dev)
    qres = torch.empty((B, M.T), dtype=torch.int32, device=dev) if want_masks else None
    free = torch.empty((B, M.T, FM), dtype=torch.int32, device=dev) if want_masks else None
    check(_large_fn(M, "backward")(C.byref(M.c), C.byref(O.c), C.c_int64(B), _ptr(_req(xs, torch.float64)),
                                    _ptr(_req(us, torch.float64)), _ptr(_req(mu, torch.float64)), _ptr(k), _ptr(K),
                                    _ptr(dJ), _ptr(info), _ptr(qres), _ptr(free), _stream()))
    if (M.n, M.m) != (FN, FM):
        k, K = k[..., :M.m].contiguous(), K[..., :M.m, :M.n].contiguous()
        free = None if free is None else free[..., :M.m].contiguous()
    return dict(k=k, K=K, dJ=dJ, info=info, qp_result=qres, free=free)


def large_forward_ls(M, alphas, xs, us, k, K, store_idx=-1):
    import torch
    B, dev = xs.shape[0], xs.device
    al = _host(alphas, (-1,))
    J = torch.empty((B, len(al)), dtype=torch.float64, device=dev)
    xs_new = torch.empty_like(xs) if store_idx >= 0 else None
    us_new = torch.empty_like(us) if store_idx >= 0 else None
    check(_large_fn(M, "forward_ls")(C.byref(M.c), C.c_int64(B), C.c_int32(len(al)), al.ctypes.data_as(_dp),
                                      C.c_int32(store_idx), _ptr(_req(xs, torch.float64)),
                                      _ptr(_req(us, torch.float64)), _ptr(_req(k, torch.float64)),
                                      _ptr(_req(K, torch.float64)), _ptr(J), _ptr(xs_new), _ptr(us_new), _stream()))
    return J, xs_new, us_new


def large_solve(M, O, x0, us_init, delta=None, k=None, K=None, trace=False, workspace=None, mu=None, dJ=None):
    import torch
    B, dev = x0.shape[0], x0.device
    _check_goal_batch(M, B)
    T, (n, m) = M.T, _large_full(M)
    padded = (M.n, M.m) != (n, m)
    delta = torch.full((B,), 2.0, dtype=torch.float64, device=dev) if delta is None else delta
    k_in, K_in = k, K
    if padded:  # the kernels work on 32 x 8: zero-padded copies in, the leading entries out
        x0, us_init = pad_last(x0, n), pad_last(us_init, m)
        k = None if k is None else pad_last(k, m)
        K = None if K is None else pad_last(K, (m, n), dims=2)
    k = torch.zeros((B, T, m), dtype=torch.float64, device=dev) if k is None else k
    K = torch.zeros((B, T, m, n), dtype=torch.float64, device=dev) if K is None else K
    res = dict(xs=torch.empty((B, T + 1, n), dtype=torch.float64, device=dev),
               us=torch.empty((B, T, m), dtype=torch.float64, device=dev),
               J=torch.empty(B, dtype=torch.float64, device=dev),
               mu=torch.ones(B, dtype=torch.float64, device=dev) if mu is None else mu,
               iters=torch.empty(B, dtype=torch.int32, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev))
    need = int(_large_fn(M, "workspace_bytes")(C.byref(M.c), C.c_int64(B)))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    tr_c, tr = None, None
    if trace:
        tr, tr_c = {}, TraceC()
        for name, dt in TRACE_FIELDS:
            if dt == "f8":
                t = torch.full((O.max_iter, B), float("nan"), dtype=torch.float64, device=dev)
            else:
                t = torch.full((O.max_iter, B), -1, dtype=torch.int32, device=dev)
            tr[name] = t
            setattr(tr_c, name, t.data_ptr())
    check(_large_fn(M, "solve_batched")(
        C.byref(M.c), C.byref(O.carrying(dJ)), C.c_int64(B), _ptr(_req(x0, torch.float64)), _ptr(_req(us_init, torch.float64)),
        _ptr(_req(delta, torch.float64)), _ptr(_req(k, torch.float64)), _ptr(_req(K, torch.float64)),
        _ptr(res["xs"]), _ptr(res["us"]), _ptr(res["J"]), _ptr(res["mu"]), _ptr(res["iters"]), _ptr(res["status"]),
        C.byref(tr_c) if tr_c is not None else None, _ptr(workspace), C.c_size_t(workspace.numel()), _stream()))
    if padded:
        res["xs"], res["us"] = res["xs"][..., :M.n].contiguous(), res["us"][..., :M.m].contiguous()
        k, K = k[..., :M.m].contiguous(), K[..., :M.m, :M.n].contiguous()
        if k_in is not None:  # in/out arguments: the caller's tensors receive the gains
            k_in.copy_(k)
            K_in.copy_(K)
            k, K = k_in, K_in
    res.update(delta=delta, k=k, K=K, trace=tr, workspace=workspace)
    return res
