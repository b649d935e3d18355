#!/usr/bin/env python3
"""Benchmark of the batched control-limited iLQR solve (cart-pole swing-up, T=100, fp64).

One "step" = one call of ilqr_solve_batched over a batch of 16384 random
swing-up problems per GPU, inputs already resident in HBM: the whole update_u
loop (<= 200 iterations, every problem to its own termination) in one launch.
With N GPUs every rank solves its own 16384 problems (weak scaling: problems
are independent, no collective inside the solve) and the step ends with the
RCCL gather of costs, iteration counts and trajectories on rank 0.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 16384
T = 100
# Algorithmic HBM bytes of one iLQR iteration of one cart-pole problem when derivative
# tensors are streamed through HBM between a derivative kernel, the backward pass and
# the line search: 8 B x (2 x (T*D + n + n^2) + 2 x ((T+1)n + Tm) + 2 x T(m + mn)),
# D = 2n^2 + 2nm + n + m + m^2, n = 5, m = 1, T = 100.
D_STEP = 2 * 25 + 2 * 5 + 5 + 1 + 1
BYTES_PER_PROBLEM_ITER = 8 * (2 * (T * D_STEP + 5 + 25) + 2 * ((T + 1) * 5 + T) + 2 * T * 6)
# fp64 operations of one iteration: backward sweep (T x (4n^3 + 10mn^2 + 6m^2 n + ...)) ~ 90 kflop,
# one line-search rollout ~ T x 150 flop (+ 3 transcendental calls per step).
FLOP_BACKWARD = 90e3
FLOP_ROLLOUT = T * 150.0
HBM_PEAK_GBS = 8000.0
FP64_VECTOR_PEAK_TFLOPS = 78.6


def make_batch(B, seed):
    r = np.random.RandomState(1 + 1000 * seed)
    st = np.stack([r.uniform(-1, 1, B), r.uniform(-1, 1, B), r.uniform(-np.pi, np.pi, B), r.uniform(-1, 1, B)], 1)
    x0 = np.stack([st[:, 0], st[:, 1], np.sin(st[:, 2]), np.cos(st[:, 2]), st[:, 3]], 1)
    us = np.random.RandomState(2 + 1000 * seed).uniform(-1, 1, (B, T, 1))
    return x0, us


def cartpole_cost():
    Q = 100.0 * np.eye(5)
    Q[1, 1] = Q[4, 4] = 10.0
    return Q, np.array([[5.0]]), 100.0 * np.eye(5), np.array([0.0, 0.0, 0.0, 1.0, 0.0])


def cpu_baseline(n_problems, threads):
    """The CPU oracle (a C port of the reference algorithm) on a bounded sample."""
    from oracle import c_oracle as co
    x0, us = make_batch(B_PER_GPU, 0)
    x0, us = x0[:n_problems], us[:n_problems]
    M = co.cartpole_model()
    co.solve(M, co.Opts(), x0[:threads], us[:threads], nthreads=threads, trace=False)  # warm
    t0 = time.perf_counter()
    out = co.solve(M, co.Opts(), x0, us, nthreads=threads, trace=False)
    dt = time.perf_counter() - t0
    return dict(value=n_problems / dt, unit="solves/s", cores=threads, kind="port",
                sample="first %d of the %d problems of rank 0, C port of the reference algorithm "
                       "(oracle/ilqr_oracle.c, gcc -O2, OpenMP over problems), %.1f s wall, "
                       "mean %.1f iterations/solve" % (n_problems, B_PER_GPU, dt, float(out["iters"].mean())))


def main_large(args):
    """--workload large: the large-state configuration (synthetic n=32, m=8, T=200, box limits +-1,
    fp64 MFMA path, one problem per wavefront).  Same contract; the roofline object is flop-based."""
    import torch
    import torch.distributed as dist
    from ilqr_amd import _hip, sharding
    from ilqr_amd.synthetic import SyntheticWorkload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d needs WORLD_SIZE=%d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    B, TL, n, m = (args.batch if args.batch is not None else 8192), 200, 32, 8
    wl = SyntheticWorkload(n=n, m=m, T=TL)
    M = wl.device_model(device=dev)
    x0_h, us_h = wl.batch(B, seed_x=5 + 1000 * rank, seed_u=6 + 1000 * rank)
    x0, us0 = torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev)
    O = _hip.Opts()
    ws = None
    res = {}

    def step():
        nonlocal ws, res
        res = _hip.large_solve(M, O, x0, us0, workspace=ws)
        ws = res["workspace"]
        if world > 1:
            for name in ("J", "iters", "status", "xs", "us"):
                sharding.gather_to_root(res[name], [B] * world, dst=0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    iters = res["iters"].cpu().numpy().astype(np.int64)
    status = res["status"].cpu().numpy()
    if rank == 0:
        flop_bw = TL * (4 * n ** 3 + 10 * m * n ** 2 + 6 * m * m * n + m ** 3 / 3.0)
        flop_fw = TL * (3 * 2 * n * n + 2 * 2 * n * m + 2 * m * m)
        flops = float(iters.sum()) * (flop_bw + flop_fw)
        tf = flops / (step_ms * 1e-3) / 1e12
        line = {
            "metric": "iLQR solves/sec (batched synthetic n=32 m=8 T=200, fp64)",
            "value": world * B * args.steps / elapsed, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_iteration": elapsed / args.steps * 1e3 / max(int(iters.max()), 1), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic non-linear system n=32, m=8, T=200, box limits +-1, %d problems per GPU, "
                                   "<=200 iterations, tol 1e-4, 30-alpha schedule" % B,
                       "problems_per_gpu": B, "horizon": TL, "parallelism": "batch-sharded x%d" % world,
                       "mean_iterations_per_solve": float(iters.mean()), "max_iterations": int(iters.max()),
                       "converged_fraction": float((status == 1).mean())},
            "roofline": {"bound": "mfma", "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
                         "kernel": "k_large_solve (whole update_u loop, one problem per wavefront)",
                         "kernel_ms": step_ms,
                         "note": "algorithmic flops = problem-iterations x (T x 225 kflop backward + T x 7.3 kflop "
                                 "for the accepted rollout); fp64 MFMA peak = fp64 vector peak on MI355X"},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, min(16, os.cpu_count() or 1))
            n_cpu = 128
            from oracle import c_oracle as co  # the CPU baseline leg: the checker timed as a C port
            Mo = co.synthetic_model(T=TL)
            t1 = time.perf_counter()
            co.solve(Mo, co.Opts(), x0_h[:n_cpu], us_h[:n_cpu], nthreads=threads, trace=False)
            dt = time.perf_counter() - t1
            line["cpu_baseline"] = dict(value=n_cpu / dt, unit="solves/s", cores=threads, kind="port",
                                        sample="first %d problems of rank 0, C port of the reference algorithm "
                                               "(OpenMP over problems), %.1f s wall" % (n_cpu, dt))
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("cartpole", "large"), default="cartpole",
                    help="cartpole: the benchmark configuration (default); large: synthetic n=32, m=8, T=200")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None,
                    help="problems per GPU (default: %d cart-pole problems, 8192 large-state problems)" % B_PER_GPU)
    ap.add_argument("--cpu-sample", type=int, default=B_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if os.environ.get("ILQR_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ILQR_BENCH_WATCHDOG"]), exit=True)
    if args.workload == "large":
        return main_large(args)

    import torch
    import torch.distributed as dist
    from ilqr_amd import _hip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d needs WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (args.gpus, args.gpus))
    # ILQR_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks
    # (ranks share devices; RCCL refuses that).  The measured configuration is always nccl (= RCCL).
    backend = os.environ.get("ILQR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B = args.batch if args.batch is not None else B_PER_GPU
    Q, R, Qt, xg = cartpole_cost()
    M = _hip.Model(T, Q, R, Qt, xg, None, [-5.0], [5.0], dt=0.05)
    O = _hip.Opts()
    x0_h, us_h = make_batch(B, rank)
    x0 = torch.tensor(x0_h, device=dev)
    us0 = torch.tensor(us_h, device=dev)
    ws = torch.empty(_hip.solve_workspace_bytes(M, B), dtype=torch.uint8, device=dev)
    # Result tensors: with more than one rank two sets alternate, so that the gather of one solve's
    # results (on the collective's stream) overlaps the next solve.
    n_sets = 2 if world > 1 else 1
    out_sets = [dict(xs=torch.empty((B, T + 1, 5), dtype=torch.float64, device=dev),
                     us=torch.empty((B, T, 1), dtype=torch.float64, device=dev),
                     J=torch.empty(B, dtype=torch.float64, device=dev),
                     mu=torch.empty(B, dtype=torch.float64, device=dev),
                     iters=torch.empty(B, dtype=torch.int32, device=dev),
                     status=torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(n_sets)]
    delta = torch.empty(B, dtype=torch.float64, device=dev)
    k = torch.empty((B, T, 1), dtype=torch.float64, device=dev)
    K = torch.empty((B, T, 1, 5), dtype=torch.float64, device=dev)
    from ilqr_amd import compaction, sharding
    cws = [None]  # workspace of the chunked solve, kept across steps
    pipe = sharding.GatherPipeline(depth=n_sets, dst=0) if world > 1 else None
    n_steps = [0]

    def step():
        slot = n_steps[0] % n_sets
        n_steps[0] += 1
        out_buf = out_sets[slot]
        if pipe is not None:
            pipe.reclaim(slot)  # the set's previous gather is ordered before the solve that overwrites it
        # a freshly constructed solver per step: delta_ = 2, k_ = K_ = 0 (ilqr_core.py:58-65)
        delta.fill_(2.0)
        k.zero_()
        K.zero_()
        if B >= compaction.COMPACT_MIN_BATCH:  # the product's policy for batches that fill the GPU
            res = _hip.solve_chunked(M, O, x0, us0, compaction.COMPACT_CHUNK, delta=delta, k=k, K=K, workspace=cws[0])
            cws[0] = res["workspace"]
            for name in out_buf:
                out_buf[name] = res[name]
        else:
            res = _hip.solve(M, O, x0, us0, delta=delta, k=k, K=K, workspace=ws, out=out_buf)
        if pipe is not None:  # the only exchange of the path: results to rank 0 over xGMI
            if backend != "nccl":  # rehearsal only: gloo's gathers of GPU tensors hang behind a running kernel
                torch.cuda.synchronize()
            pipe.submit(slot, {name: out_buf[name] for name in ("J", "iters", "status", "xs", "us")})
        return res

    def barrier():
        if pipe is not None:
            pipe.drain()  # orders the stream behind every gather issued so far
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms_events = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    out_buf = out_sets[(n_steps[0] - 1) % n_sets]
    iters = out_buf["iters"].cpu().numpy().astype(np.int64)
    status = out_buf["status"].cpu().numpy()
    J = out_buf["J"].cpu().numpy()
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        solves_per_s = world * B * args.steps / elapsed
        prob_iters = int(iters.sum())
        max_iters = int(iters.max())
        alg_bytes = prob_iters * BYTES_PER_PROBLEM_ITER
        dur_s = step_ms_events * 1e-3
        achieved = alg_bytes / dur_s / 1e9
        flops = prob_iters * (FLOP_BACKWARD + FLOP_ROLLOUT)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("batch") == B and tj.get("kernel") == "k_solve":
                traffic = tj.get("hbm_bytes_per_launch")
        line = {
            "metric": "iLQR solves/sec (batched cartpole T=100, fp64)",
            "value": solves_per_s,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_iteration": ms_per_step / max(max_iters, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "batched cartpole swing-up, %d random initial states per GPU, T=100, n=5, m=1, "
                                   "control limits +-5 active, <=200 iterations, tol 1e-4, 30-alpha schedule" % B,
                       "problems_per_gpu": B, "horizon": T, "parallelism": "batch-sharded x%d" % world,
                       "mean_iterations_per_solve": float(iters.mean()), "max_iterations": max_iters,
                       "converged_fraction": float((status == 1).mean()), "mean_final_cost": float(J.mean())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("k_solve (whole update_u loop, one launch per step; its pack/unpack "
                                    "transposes are inside the timed span)" if B < compaction.COMPACT_MIN_BATCH else
                                    "k_solve in launches of %d iterations, finished problems dropped in between "
                                    "(gathers and transposes inside the timed span)" % compaction.COMPACT_CHUNK),
                         "kernel_ms": step_ms_events,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "algorithmic bytes = problem-iterations x %d B (derivative tensors streamed "
                                 "through HBM); the kernel fuses derivative evaluation into the sweeps, so its "
                                 "real HBM traffic is far lower and the binding resource is FP64 VALU issue"
                                 % BYTES_PER_PROBLEM_ITER},
            "fp64_valu": {"achieved_tflops": flops / dur_s / 1e12, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                          "frac": flops / dur_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                          "flop_model": "problem-iterations x (90 kflop backward + 15 kflop per rollout)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, min(16, os.cpu_count() or 1))
            line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, B), threads)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
