"""User-defined dynamics on the GPU path (up to 8 states, 1 to 4 controls).

The reference takes any Python object with f / f_x / f_u (``AutogradDynamics``,
``FiniteDifferenceDynamics``, ilqr/ilqr_dynamics.py:94-575).  Python callbacks cannot run inside a
kernel; here the model is a small C source (see include/ilqr_user.h for the contract) and the
solver kernels are compiled around it with hipcc, once per model, into
``ilqr_amd/libilqr_user_<name>.so``.  ``UserDynamics`` wraps that library as a ``Dynamics`` that
``iLQR`` accepts.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _hip
from .ilqr_dynamics import Dynamics

_HERE = os.path.dirname(os.path.abspath(__file__))
_TU = os.path.join(_HERE, "csrc", "ilqr_user_tu.hip")
_DEPS = [_TU, os.path.join(_HERE, "csrc", "ilqr_device.hpp"), os.path.join(_HERE, "csrc", "ilqr_math.hpp"),
         os.path.join(_HERE, "csrc", "ilqr_qp_dense.hpp"),
         os.path.join(os.path.dirname(_HERE), "include", "ilqr_user.h"),
         os.path.join(os.path.dirname(_HERE), "include", "ilqr_user_fd.h"),
         os.path.join(os.path.dirname(_HERE), "include", "ilqr_hip.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


class UserModelC(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("T", C.c_int32), ("n_params", C.c_int32),
                ("cost_kind", C.c_int32), ("pad_", C.c_int32)] + \
               [(k, C.POINTER(C.c_double)) for k in ("params", "Q", "R", "Q_terminal", "x_goal", "u_goal", "lower",
                                                     "upper")] + \
               [("goal_stride_x", C.c_int64), ("goal_stride_u", C.c_int64)]


def library_path(name):
    return os.path.join(_HERE, "libilqr_user_%s.so" % name)


def build_user_model(header, name, force=False):
    """Compile the solver around the model header ``header`` (hipcc, gfx950; needs no GPU)."""
    header = os.path.abspath(header)
    out = library_path(name)
    if (not force and os.path.exists(out)
            and all(os.path.getmtime(out) >= os.path.getmtime(f) for f in _DEPS + [header])):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc] + HIPCC_FLAGS + ["-DILQR_USER_MODEL=\"%s\"" % header, _TU, "-o", out], check=True)
    return out


class UserModel(object):
    """ctypes view of one per-model library + the host arrays of one problem family."""

    def __init__(self, lib, T, params, Q, R, Q_terminal=None, x_goal=None, u_goal=None, lower=None, upper=None,
                 x_path=None, u_path=None):
        """QRCost: x_goal [n], u_goal [m] (host).  PathQRCost: x_path [T+1, n], u_path [T, m] as device
        tensors (time-indexed goals shared by the batch; the value then carries no 1/2).
        One goal or path per problem: x_goal [B, n], u_goal [B, m] or x_path [B, T+1, n], u_path [B, T, m]."""
        n, npar = C.c_int32(), C.c_int32()
        _hip.check(lib.ilqr_user_dims(C.byref(n), C.byref(npar)))
        m = int(lib.ilqr_user_controls())
        self.lib, self.n, self.m, self.T = lib, n.value, m, int(T)
        host = _hip._host
        dp = C.POINTER(C.c_double)
        self.batch_goals = None
        per_problem_goal = x_path is None and x_goal is not None and _hip._ndim(x_goal) == 2
        self.a = dict(params=host(params, (npar.value,)), Q=host(Q, (self.n, self.n)), R=host(R, (m, m)),
                      Q_terminal=host(Q if Q_terminal is None else Q_terminal, (self.n, self.n)),
                      x_goal=host(np.zeros(self.n) if (x_goal is None or per_problem_goal) else x_goal, (self.n,)),
                      u_goal=host(np.zeros(m) if (u_goal is None or per_problem_goal) else u_goal, (m,)),
                      lower=host(lower, (m,)), upper=host(upper, (m,)))
        c = UserModelC()
        c.n, c.m, c.T, c.n_params = self.n, m, int(T), npar.value
        for k, v in self.a.items():
            setattr(c, k, v.ctypes.data_as(dp))
        c.cost_kind = _hip.COST_QR
        self.gx = self.gu = None  # device goals (paths and / or per-problem goals)
        if x_path is not None or u_path is not None:
            import torch
            if x_path is None or u_path is None:
                raise ValueError("PathQRCost needs both x_path and u_path")
            if _hip._ndim(x_path) == 3:
                self.gx, self.gu = _hip._dev64(x_path), _hip._dev64(u_path)
                Bg = self.gx.shape[0]
                if tuple(self.gx.shape) != (Bg, int(T) + 1, self.n) or tuple(self.gu.shape) != (Bg, int(T), m):
                    raise ValueError("per-problem paths must be x_path [B, %d, %d] and u_path [B, %d, %d]"
                                     % (int(T) + 1, self.n, int(T), m))
                self.batch_goals = int(Bg)
                c.goal_stride_x, c.goal_stride_u = (int(T) + 1) * self.n, int(T) * m
            else:
                self.gx = _hip._req(x_path, torch.float64)
                self.gu = _hip._req(u_path, torch.float64)
                if tuple(self.gx.shape) != (int(T) + 1, self.n) or tuple(self.gu.shape) != (int(T), m):
                    raise ValueError("x_path must be [%d, %d] and u_path [%d, %d]" % (int(T) + 1, self.n, int(T), m))
            c.cost_kind = _hip.COST_PATH_QR
        elif per_problem_goal:
            self.gx = _hip._dev64(x_goal)
            Bg = self.gx.shape[0]
            self.gu = _hip._dev64(np.zeros((Bg, m)) if u_goal is None else u_goal)
            if tuple(self.gx.shape) != (Bg, self.n) or tuple(self.gu.shape) != (Bg, m):
                raise ValueError("per-problem goals must be x_goal [B, %d] and u_goal [B, %d]" % (self.n, m))
            self.batch_goals = int(Bg)
            c.goal_stride_x, c.goal_stride_u = self.n, m
        if self.gx is not None:
            c.x_goal = C.cast(C.c_void_p(self.gx.data_ptr()), dp)
            c.u_goal = C.cast(C.c_void_p(self.gu.data_ptr()), dp)
        self.x_path, self.u_path = self.gx, self.gu
        self.c = c

    def subset(self, idx):
        """The model for the sub-batch ``idx`` of the problems it was built for (per-problem goals gathered)."""
        if self.batch_goals is None:
            return self
        import copy
        other = copy.copy(self)
        other.gx, other.gu = self.gx[idx].contiguous(), self.gu[idx].contiguous()
        other.x_path, other.u_path = other.gx, other.gu
        other.batch_goals = int(other.gx.shape[0])
        c = UserModelC()
        C.memmove(C.byref(c), C.byref(self.c), C.sizeof(UserModelC))
        dp = C.POINTER(C.c_double)
        c.x_goal = C.cast(C.c_void_p(other.gx.data_ptr()), dp)
        c.u_goal = C.cast(C.c_void_p(other.gu.data_ptr()), dp)
        other.c = c
        return other


# ---- models with more than 8 states: the large-state path compiled around the same kind of header

_LARGE_TU = os.path.join(_HERE, "csrc", "ilqr_large.hip")


def large_library_path(name):
    return os.path.join(_HERE, "libilqr_largeuser_%s.so" % name)


def build_large_user_model(header, name, force=False):
    """Compile the large-state solver (one problem per wavefront, up to 32 states and 8 controls) around the
    model header ``header`` (same contract as ``build_user_model``; hipcc, gfx950; needs no GPU)."""
    header = os.path.abspath(header)
    out = large_library_path(name)
    deps = [_LARGE_TU, header, os.path.join(_HERE, "csrc", "ilqr_math.hpp"),
            os.path.join(_HERE, "..", "include", "ilqr_hip.h"), os.path.join(_HERE, "..", "include", "ilqr_user_fd.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(f) for f in deps):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc] + HIPCC_FLAGS + ["-DILQR_LARGE_USER_MODEL=\"%s\"" % header, _LARGE_TU, "-o", out], check=True)
    return out


def load_large(name):
    path = large_library_path(name)
    if not os.path.exists(path):
        raise _hip.IlqrHipError("%s is missing: build it with ilqr_amd.user_model.build_large_user_model(header, %r)"
                                % (path, name))
    lib = C.CDLL(path)
    lib.ilqr_large_workspace_bytes.restype = C.c_size_t
    return lib


def large_user_dims(lib):
    n, m, npar = C.c_int32(), C.c_int32(), C.c_int32()
    _hip.check(lib.ilqr_large_user_dims(C.byref(n), C.byref(m), C.byref(npar)))
    return n.value, m.value, npar.value


def LargeUserModel(lib, T, params, Q, R, Q_terminal=None, x_goal=None, u_goal=None, lower=None, upper=None,
                   x_path=None, u_path=None, device=None):
    """The device model (an ``_hip.LargeModel``) of a library built by ``build_large_user_model``: every
    ``_hip.large_*`` wrapper then runs that library's kernels."""
    n, m, npar = large_user_dims(lib)
    params = np.asarray(params, dtype=np.float64).reshape(npar)
    return _hip.LargeModel(T, None, None, None, Q, R, Q_terminal, x_goal, u_goal, lower, upper, device=device,
                           x_path=x_path, u_path=u_path, user=(lib, params, n, m))


class LargeUserDynamics(Dynamics):
    """A ``Dynamics`` with more than 8 states (up to 32, with up to 8 controls) backed by a per-model library of the
    large-state path (``build_large_user_model(header, name)``); arguments as ``UserDynamics``."""

    device_model = "large_user"

    def __init__(self, name, params, max_control):
        self.name = name
        self.lib = load_large(name)
        self._n, self._m, npar = large_user_dims(self.lib)
        self.params = np.asarray(params, dtype=np.float64).reshape(npar)
        self.upper_bound = np.broadcast_to(np.asarray(max_control, dtype=np.float64), (self._m,)).copy()
        self.lower_bound = -self.upper_bound

    state_dimension = property(lambda self: self._n)
    control_dimension = property(lambda self: self._m)
    has_hessians = property(lambda self: False)

    def f(self, x, u, i):
        raise NotImplementedError("point evaluations of a user model are not exported by its library")

    f_x = f_u = f


def load(name):
    path = library_path(name)
    if not os.path.exists(path):
        raise _hip.IlqrHipError("%s is missing: build it with ilqr_amd.user_model.build_user_model(header, %r)"
                                % (path, name))
    lib = C.CDLL(path)
    lib.ilqr_user_controls.restype = C.c_int
    lib.ilqr_user_workspace_bytes.restype = C.c_size_t
    lib.ilqr_user_workspace_bytes.argtypes = [C.POINTER(UserModelC), C.c_int64]
    return lib


def solve(M, O, x0, us_init, delta=None, k=None, K=None, trace=False, workspace=None, mu=None, dJ=None):
    """ilqr_user_solve_batched on reference-layout device tensors (same dict as _hip.solve)."""
    import torch
    B, dev = x0.shape[0], x0.device
    _hip._check_goal_batch(M, B)
    T, n, m = M.T, M.n, M.m
    f64 = torch.float64
    delta = torch.full((B,), 2.0, dtype=f64, device=dev) if delta is None else delta
    k = torch.zeros((B, T, m), dtype=f64, device=dev) if k is None else k
    K = torch.zeros((B, T, m, n), dtype=f64, device=dev) if K is None else K
    res = dict(xs=torch.empty((B, T + 1, n), dtype=f64, device=dev), us=torch.empty((B, T, m), dtype=f64, device=dev),
               J=torch.empty(B, dtype=f64, device=dev), mu=torch.ones(B, dtype=f64, device=dev) if mu is None else mu,
               iters=torch.empty(B, dtype=torch.int32, device=dev), status=torch.empty(B, dtype=torch.int32, device=dev))
    need = int(M.lib.ilqr_user_workspace_bytes(C.byref(M.c), C.c_int64(B)))
    if workspace is None or workspace.numel() < need:  # (the need is not monotone in B: fewer problems per
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)  # wavefront below 32768 problems)
    tr_c, tr = None, None
    if trace:
        tr, tr_c = {}, _hip.TraceC()
        for name, dt in _hip.TRACE_FIELDS:
            t = (torch.full((O.max_iter, B), float("nan"), dtype=f64, device=dev) if dt == "f8"
                 else torch.full((O.max_iter, B), -1, dtype=torch.int32, device=dev))
            tr[name] = t
            setattr(tr_c, name, t.data_ptr())
    p, req = _hip._ptr, _hip._req
    _hip.check(M.lib.ilqr_user_solve_batched(
        C.byref(M.c), C.byref(O.carrying(dJ)), C.c_int64(B), p(req(x0.reshape(B, n), f64)), p(req(us_init.reshape(B, T, m), f64)),
        p(req(delta, f64)), p(req(k, f64)), p(req(K, f64)), p(res["xs"]), p(res["us"]), p(res["J"]), p(res["mu"]),
        p(res["iters"]), p(res["status"]), C.byref(tr_c) if tr_c is not None else None, p(workspace),
        C.c_size_t(workspace.numel()), _hip._stream()))
    res.update(delta=delta, k=k, K=K, trace=tr, workspace=workspace)
    return res


class UserDynamics(Dynamics):
    """A ``Dynamics`` backed by a per-model device library.

    name: the library built by ``build_user_model(header, name)``; params: the model's parameter
    vector p; max_control: |u| <= max_control, a number or one per control (``lower_bound`` /
    ``upper_bound``, as CartPoleDynamics)."""

    device_model = "user"

    def __init__(self, name, params, max_control):
        self.name = name
        self.lib = load(name)
        n, npar = C.c_int32(), C.c_int32()
        _hip.check(self.lib.ilqr_user_dims(C.byref(n), C.byref(npar)))
        self._n, self._m = n.value, int(self.lib.ilqr_user_controls())
        self.params = np.asarray(params, dtype=np.float64).reshape(npar.value)
        self.upper_bound = np.broadcast_to(np.asarray(max_control, dtype=np.float64), (self._m,)).copy()
        self.lower_bound = -self.upper_bound

    @property
    def state_dimension(self):
        return self._n

    @property
    def control_dimension(self):
        return self._m

    @property
    def has_hessians(self):
        return False

    def f(self, x, u, i):
        raise NotImplementedError("point evaluations of a user model are not exported by its library")

    f_x = f_u = f
