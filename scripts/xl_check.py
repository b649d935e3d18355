"""Stage-by-stage comparison of the LDS-resident (up to 64 x 16) path with the oracle (diagnostic)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip as hip
from oracle import c_oracle as co

n, m = int(sys.argv[1]), int(sys.argv[2])
T, B, n_it = int(sys.argv[3]) if len(sys.argv) > 3 else 40, int(sys.argv[4]) if len(sys.argv) > 4 else 24, 15
dev = lambda a: torch.tensor(a, device="cuda:0")
eq = lambda a, b: np.array_equal(a, b)
Mo = co.synthetic_model(n=n, m=m, T=T, seed=40 + n + m, u_max=0.6)
M = hip.LargeModel(T, Mo.A, Mo.B, Mo.W, Mo.Q, Mo.R, Mo.Qt, Mo.x_goal, Mo.u_goal, Mo.lower, Mo.upper, dt=Mo.dt, sig_c=Mo.sig_c)
print("xl:", M.xl, M.full)
x0, us0 = co.synthetic_batch(B, n=n, m=m, T=T, seed_x=50 + n, seed_u=60 + m)
us0 *= 2.0
xs_ref, J_ref = co.rollout_cost(Mo, x0, us0)
xs, J = hip.large_rollout_cost(M, dev(x0), dev(us0))
torch.cuda.synchronize()
print("rollout xs", eq(xs.cpu().numpy(), xs_ref), "J", eq(J.cpu().numpy(), J_ref), np.abs(J.cpu().numpy() - J_ref).max())
bw_ref = co.backward(Mo, co.Opts(), xs_ref, us0, np.ones(B))
bw = hip.large_backward(M, hip.Opts(), dev(xs_ref), dev(us0), dev(np.ones(B)))
torch.cuda.synchronize()
info = bw["info"].cpu().numpy()
print("backward rc", info[4][:8], "stopped", info[0][:8], bw_ref["stopped_at"][:8])
for key in ("k", "K"):
    a, b = bw[key].cpu().numpy(), bw_ref[key]
    print(" ", key, eq(a, b), "max diff", np.abs(a - b).max(), "first bad step (from the end)",
          None if eq(a, b) else max(np.argwhere((a != b).reshape(B, T, -1).any(-1))[:, 1]))
print("  qp_result", eq(bw["qp_result"].cpu().numpy(), bw_ref["qp_result"]), "free", eq(bw["free"].cpu().numpy(), bw_ref["free"]))
print("  qp_iter", eq(info[1], bw_ref["qp_iter"]), "backtrack", eq(info[2], bw_ref["qp_backtrack"]), "n_clamped", eq(info[3], bw_ref["n_clamped"]))
dJ = bw["dJ"].cpu().numpy()
print("  dJ", eq(dJ[0], bw_ref["dJ1"]), eq(dJ[1], bw_ref["dJ2"]))
al = co.Opts().alphas[:9]
Jf_ref = co.forward(Mo, al, xs_ref, us0, bw_ref["k"], bw_ref["K"])
Jf, xn, un = hip.large_forward_ls(M, al, dev(xs_ref), dev(us0), dev(bw_ref["k"]), dev(bw_ref["K"]), store_idx=2)
torch.cuda.synchronize()
print("forward J", eq(Jf.cpu().numpy(), Jf_ref[2]), "xs", eq(xn.cpu().numpy(), Jf_ref[0][:, 2]), "us", eq(un.cpu().numpy(), Jf_ref[1][:, 2]))
t = time.time()
ref = co.solve(Mo, co.Opts(max_iter=n_it), x0, us0, nthreads=8)
print("oracle solve %.1f s" % (time.time() - t))
out = hip.large_solve(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0), trace=True)
torch.cuda.synchronize()
for key in ("xs", "us", "J", "mu", "iters", "status", "k", "K", "delta"):
    print("solve", key, eq(out[key].cpu().numpy(), ref[key]))
for key in ("J", "accepted", "alpha_idx", "mu", "delta", "qp_iter", "qp_backtrack", "n_clamped", "stopped_at"):
    a, b = out["trace"][key].cpu().numpy().T, ref["trace"][key]
    ok = np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(np.where(b < 0, a, a), b) or np.array_equal(a[b >= -1], b[b >= -1])
    print("trace", key, ok)
print("iters", ref["iters"][:12], "clamped steps", (ref["trace"]["n_clamped"] > 0).any())
t0 = time.time()
for _ in range(3):
    out = hip.large_solve(M, hip.Opts(max_iter=n_it), dev(x0), dev(us0))
torch.cuda.synchronize()
print("gpu solve %.1f ms" % ((time.time() - t0) / 3 * 1e3))
