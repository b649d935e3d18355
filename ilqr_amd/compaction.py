"""Solving a large batch in chunks of iterations, dropping finished problems between chunks.

``update_u`` (ilqr/ilqr_core.py:241-399) ends per problem: on convergence, on mu >= mu_max, or after
the iteration budget.  In one launch a lane whose problem has ended idles until the slowest problem
of its wavefront ends (on the swing-up benchmark the mean is 117 of 200 iterations), and with every
SIMD busy those idle lanes cost real time and memory traffic.  Here the solve is cut into launches of
``chunk`` iterations; after each one the problems that are still running (status "ran out of
iterations") are gathered into a dense sub-batch and continued from their carried state: controls,
mu, delta and the gains k, K.  A continued launch re-rolls the current controls (same inputs, same
bits, same cost) and proceeds exactly as the single launch would have: the iterates, iteration
counts and results are identical (the per-iteration callback mode of ``iLQR.update_u`` relies on the
same property with chunks of one iteration).  The expected-improvement sums Delta_J1 / Delta_J2 of
the last backward pass travel with the problem too (``dJ_carry`` of the C options): the reference
re-uses them when a backward pass gives up on a non-finite Q_uu at mu_max (ilqr_core.py:291-305).
For the built-in cart-pole path the same loop exists behind the C ABI
(``ilqr_solve_batched_chunked``); this module serves the per-model libraries and callers that want
the per-iteration trace.

Worth it only when the batch fills the GPU (tens of thousands of problems); below that a launch is
as long as its slowest problem whatever the others do.
"""
import numpy as np

from . import _hip

#: problems from which ``solve_auto`` switches to chunked launches, and the chunk length
COMPACT_MIN_BATCH = 49152
COMPACT_CHUNK = 32


#: problems in one round of wavefronts on an MI355X: 256 CUs x 4 SIMDs, one 64-problem wavefront each
ROUND_PROBLEMS = 256 * 4 * 64


def solve_compacting(solve_fn, M, opts_kw, max_iter, x0, us_init, delta=None, k=None, K=None, chunk=COMPACT_CHUNK,
                     trace=False):
    """``solve_fn``: _hip.solve or user_model.solve (same keywords).  ``opts_kw``: keyword arguments of
    ``_hip.Opts`` except max_iter / mu_from_array.  Returns the dict of ``_hip.solve``."""
    import torch
    B, dev = x0.shape[0], x0.device
    T, n, m = M.T, M.n, M.m
    f64 = torch.float64
    delta = torch.full((B,), 2.0, dtype=f64, device=dev) if delta is None else delta
    k = torch.zeros((B, T, m), dtype=f64, device=dev) if k is None else k
    K = torch.zeros((B, T, m, n), dtype=f64, device=dev) if K is None else K
    mu = torch.ones(B, dtype=f64, device=dev)
    dJ = torch.zeros((2, B), dtype=f64, device=dev)  # Delta_J1, Delta_J2 of each problem's last backward pass
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    status = torch.full((B,), _hip.STATUS_RAN_OUT, dtype=torch.int32, device=dev)
    x0 = x0.reshape(B, n)
    us = us_init.reshape(B, T, m)
    xs = J = None
    tr = None
    if trace:
        tr = {name: (torch.full((max_iter, B), float("nan"), dtype=f64, device=dev) if dt == "f8"
                     else torch.full((max_iter, B), -1, dtype=torch.int32, device=dev))
              for name, dt in _hip.TRACE_FIELDS}
    opts = {}
    ws = None
    idx = None  # None: every problem is still running and the launches work on the full arrays
    done = 0
    while True:
        n_run = B if idx is None else int(idx.numel())
        # slightly more problems than a whole number of rounds of wavefronts holds: shorter chunks, so that the
        # running set is counted again sooner and drops a round earlier (the overhang costs most of a round)
        overhang = n_run > ROUND_PROBLEMS and 0 < n_run % ROUND_PROBLEMS <= ROUND_PROBLEMS // 6
        n_it = min(max(1, chunk // 4) if overhang else chunk, max_iter - done)
        if n_it not in opts:
            opts[n_it] = _hip.Opts(max_iter=n_it, mu_from_array=True, **opts_kw)
        if idx is None:
            res = solve_fn(M, opts[n_it], x0, us, delta=delta, k=k, K=K, mu=mu, trace=trace, workspace=ws, dJ=dJ)
            ws = res["workspace"]
            xs, us, J = res["xs"], res["us"], res["J"]
            iters += res["iters"]
            status = res["status"]
        else:
            d_i, k_i, K_i, mu_i, dJ_i = delta[idx], k[idx], K[idx], mu[idx], dJ[:, idx].contiguous()
            M_i = M.subset(idx) if hasattr(M, "subset") else M  # (per-problem goals follow their problems)
            res = solve_fn(M_i, opts[n_it], x0[idx], us[idx], delta=d_i, k=k_i, K=K_i, mu=mu_i, trace=trace, workspace=ws,
                           dJ=dJ_i)
            delta[idx], k[idx], K[idx], mu[idx], dJ[:, idx] = d_i, k_i, K_i, mu_i, dJ_i
            xs[idx], us[idx], J[idx] = res["xs"], res["us"], res["J"]
            iters[idx] += res["iters"]
            status[idx] = res["status"]
        if trace and n_it > 0:
            for name in tr:
                if idx is None:
                    tr[name][done:done + n_it] = res["trace"][name]
                else:
                    tr[name][done:done + n_it, idx] = res["trace"][name]
        done += n_it
        if done >= max_iter:
            break
        running = res["status"] == _hip.STATUS_RAN_OUT
        if idx is None:
            if bool(running.all()):
                continue
            idx = torch.nonzero(running).flatten()
        else:
            idx = idx[running]
        if idx.numel() == 0:
            break
    return dict(xs=xs, us=us, J=J, mu=mu, iters=iters, status=status, delta=delta, k=k, K=K, trace=tr, workspace=ws)


def solve_auto(solve_fn, M, opts_kw, max_iter, x0, us_init, delta=None, k=None, K=None, trace=False):
    """One launch for batches that do not fill the GPU, chunked launches with compaction above."""
    if x0.shape[0] >= COMPACT_MIN_BATCH and max_iter > 2 * COMPACT_CHUNK:
        if isinstance(M, _hip.Model) and not trace:  # the built-in path has the same loop behind the C ABI
            return _hip.solve_chunked(M, _hip.Opts(max_iter=max_iter, **opts_kw), x0, us_init, COMPACT_CHUNK,
                                      delta=delta, k=k, K=K)
        return solve_compacting(solve_fn, M, opts_kw, max_iter, x0, us_init, delta=delta, k=k, K=K, trace=trace)
    return solve_fn(M, _hip.Opts(max_iter=max_iter, **opts_kw), x0, us_init, delta=delta, k=k, K=K, trace=trace)
