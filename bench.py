#!/usr/bin/env python3
"""Benchmark of the batched control-limited iLQR solve (cart-pole swing-up, T=100, fp64).

One "step" = one solve of this rank's batch of random swing-up problems, inputs already
resident in HBM: the whole update_u loop (<= 200 iterations, every problem to its own
termination) on the device, one launch up to 32768 problems per GPU and launches of 32
iterations with the finished problems dropped in between above that.

  python bench.py                        16384 problems on one GPU (the benchmark configuration)
  ... --gpus N                           weak scaling: every rank solves its own 16384 problems
  ... --gpus N --total 1048576           strong scaling: the million-problem configuration, the batch cut
                                         into contiguous slices with ilqr_amd.sharding.shard_bounds
Problems are independent, so there is no collective inside the solve; with N > 1 every step ends with
the gather of costs, iteration counts and trajectories on rank 0 (RCCL over xGMI), overlapped with the
next step's solve.

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches itself: the parent (which never
touches a GPU) starts N child processes, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT set, relays rank 0's JSON line and exits non-zero if any child does.  Under an external launcher
(torch.distributed.run sets WORLD_SIZE) it runs as one rank of that job.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 16384
T = 100
N_STATE, N_CTRL = 5, 1
# HBM bytes of one iLQR iteration of one cart-pole problem.
#  * staged design (derivative tensors streamed through HBM between a derivative kernel, the backward pass and
#    the line search): 8 B x (2 x (T*D + n + n^2) + 2 x ((T+1)n + Tm) + 2 x T(m + mn)), D = 2n^2 + 2nm + n + m + m^2
#  * fused design (what k_solve does: derivatives evaluated in registers): the backward sweep reads xs, us and
#    writes k, K; the accepted rollout reads xs, us, k, K and writes xs, us
D_STEP = 2 * 25 + 2 * 5 + 5 + 1 + 1
STAGED_BYTES_PER_PROBLEM_ITER = 8 * (2 * (T * D_STEP + 5 + 25) + 2 * ((T + 1) * 5 + T) + 2 * T * 6)
FUSED_READ_BYTES = 8 * (((T + 1) * 5 + T) + ((T + 1) * 5 + T + T * 6))
FUSED_WRITE_BYTES = 8 * (T * 6 + ((T + 1) * 5 + T))
# fp64 operations of one iteration: backward sweep (T x (4n^3 + 10mn^2 + 6m^2 n + ...)) ~ 90 kflop,
# one line-search rollout ~ T x 150 flop (+ 3 transcendental calls per step).
FLOP_BACKWARD = 90e3
FLOP_ROLLOUT = T * 150.0
HBM_PEAK_GBS = 8000.0
FP64_VECTOR_PEAK_TFLOPS = 78.6
N_SIMD = 1024  # 256 CUs x 4
# The reference as written (its own iLQR.update_u, Python, dynamics through a sympy-lambdified stand-in for the
# Theano model) on the single-problem example, timed once in the build container (one Xeon core): it cannot
# travel to the GPU box, so the number is a recorded constant, not a measurement of this run.
REFERENCE_AS_WRITTEN = {"value": 0.07, "unit": "solves/s", "cores": 1,
                        "sample": "examples/cartpole_example.py configuration, 200 iterations in 13.9 s "
                                  "(69 ms per iteration), CPython 3.10, one Xeon core of the build container; "
                                  "recorded, not measured by this run"}


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
                       "(oracle/ilqr_oracle.c: gcc -O2 -ffp-contract=off, OpenMP over problems, its own "
                       "deterministic sin/cos/atan2 instead of libm so that results equal the GPU's bit for "
                       "bit), %.1f s wall, mean %.1f iterations/solve"
                       % (n_problems, B_PER_GPU, dt, float(out["iters"].mean())),
                reference_as_written=REFERENCE_AS_WRITTEN)


def self_launch(n_ranks, argv):
    """Parent of a --gpus N run started without a launcher: one child per rank, rank 0's stdout relayed.
    Returns the exit status for the parent.  Nothing here touches the GPU (no torch import)."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    children = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                         stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    status = 0
    try:
        pending = set(range(n_ranks))
        out0 = b""
        while pending:
            for r in sorted(pending):
                rc = children[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if r == 0:
                    out0 = children[0].stdout.read()
                if rc != 0:
                    sys.stderr.write("bench.py: rank %d exited with status %d\n" % (r, rc))
                    status = status or (rc if rc > 0 else 1)
            if status:
                break
            if pending:
                time.sleep(0.05)
    finally:
        for c in children:  # a failed rank leaves its peers waiting in a collective: end exactly those processes
            if c.poll() is None:
                c.terminate()
        for c in children:
            try:
                c.wait(timeout=20)
            except subprocess.TimeoutExpired:
                c.kill()
    if status == 0:
        lines = [ln for ln in out0.decode().splitlines() if ln.startswith("{")]
        if len(lines) != 1:
            sys.stderr.write("bench.py: rank 0 printed %d JSON lines\n" % len(lines))
            return 1
        print(lines[0])
    return status


class ShardedSolveJob(object):
    """The benchmark step of one rank: solve the rank's problems into one of ``n_sets`` sets of result
    tensors, then hand that set to the gather pipeline (N > 1), which moves it to rank 0 while the next
    step's solve runs into the other set.

    solve_into(out): runs the solve, writing xs, us, J, iters, status (and whatever else it likes) into
        the tensors of the dict ``out`` -- the same tensors every time the set comes round again.
    out_sets: the result sets (dicts of preallocated tensors, leading dimension = problems of this rank).
    pipe: ilqr_amd.sharding.GatherPipeline or None (single rank).
    before_submit: called between the solve and the gather (the gloo rehearsal synchronises the device
        there; RCCL orders the gather behind the solve on the stream by itself).
    """
    GATHERED = ("J", "iters", "status", "xs", "us")

    def __init__(self, solve_into, out_sets, pipe=None, before_submit=None, mark=None):
        self.solve_into, self.out_sets, self.pipe, self.before_submit = solve_into, out_sets, pipe, before_submit
        self.n_steps = 0
        # mark(): a point in the stream's time; taken after every solve and after the gathers have been drained,
        # so that the part of the last gather that no solve hides can be reported
        self.mark, self.solve_end, self.drained = mark, None, None

    def step(self):
        slot = self.n_steps % len(self.out_sets)
        self.n_steps += 1
        out = self.out_sets[slot]
        if self.pipe is not None:
            self.pipe.reclaim(slot)  # the set's previous gather is ordered before the solve that overwrites it
        self.solve_into(out)
        if self.mark is not None:
            self.solve_end = self.mark()
        if self.pipe is not None:  # the only exchange of the path: results to rank 0
            if self.before_submit is not None:
                self.before_submit()
            self.pipe.submit(slot, {name: out[name] for name in self.GATHERED})
        return slot

    def finish(self):
        if self.pipe is not None:
            self.pipe.drain()  # orders the stream behind every gather issued so far
        if self.mark is not None:
            self.drained = self.mark()

    def last_slot(self):
        return (self.n_steps - 1) % len(self.out_sets)


def run_steps(job, steps, warmup, barrier, record=None):
    """W untimed steps, then exactly K timed ones between two barriers.  Returns the wall time of the K
    steps on this rank.  ``record(i, "start" | "end")`` brackets every timed step (device events)."""
    for _ in range(warmup):
        job.step()
    job.finish()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        if record is not None:
            record(i, "start")
        job.step()
        if record is not None:
            record(i, "end")
    job.finish()
    barrier()
    return time.perf_counter() - t0


def sector_granular_factor(iters, group=8):
    """HBM moves 64-byte sectors = 8 consecutive problems of one row of the batch-minor workspace, and a
    sector is read or written while ANY of its 8 problems is still iterating.  Returns
    (sum over groups of 8 of their longest solve x 8) / (sum of the problems' iterations): the factor by
    which memory traffic exceeds the per-problem byte count."""
    it = np.asarray(iters, dtype=np.int64)
    pad = (-len(it)) % group
    if pad:
        it = np.concatenate([it, np.zeros(pad, dtype=np.int64)])
    return float(it.reshape(-1, group).max(1).sum() * group) / max(float(it.sum()), 1.0)


def load_counters(workload, batch):
    """The PMC-derived numbers of this workload and batch size (profiles/solve_counters.json, written by
    scripts/summarise_profiles.py from the rocprofv3 passes named inside it), or None."""
    path = os.path.join(ROOT, "profiles", "solve_counters.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        for entry in json.load(fh).get("entries", []):
            if entry.get("workload") == workload and entry.get("batch") == batch:
                return entry
    return None


def load_critical_path():
    """profiles/critical_path.json (scripts/chain_depth.py: lengths of the loop-carried dependent instruction chains
    of one backward step and one rollout step, counted from the compiled code), or None."""
    path = os.path.join(ROOT, "profiles", "critical_path.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)


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
        raise SystemExit("bench.py --gpus %d runs under a launcher with WORLD_SIZE=%d" % (args.gpus, world))
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
        cnt = load_counters("large", B)
        line = {
            "metric": "iLQR solves/sec (batched synthetic n=32 m=8 T=200, fp64)",
            "value": world * B * args.steps / elapsed, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_iteration": elapsed / args.steps * 1e3 / max(int(iters.max()), 1), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic non-linear system n=32, m=8, T=200, box limits +-1, %d problems per GPU, "
                                   "<=200 iterations, tol 1e-4, 30-alpha schedule" % B,
                       "problems_per_gpu": B, "horizon": TL, "parallelism": "batch-sharded x%d" % world,
                       "ranks_seen": int(dist.get_world_size()) if world > 1 else 1,
                       "mean_iterations_per_solve": float(iters.mean()), "max_iterations": int(iters.max()),
                       "converged_fraction": float((status == 1).mean())},
            "roofline": {"bound": "mfma", "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / FP64_VECTOR_PEAK_TFLOPS,
                         "traffic": cnt.get("hbm_bytes_per_launch") if cnt else None,
                         "kernel": "k_large_solve (whole update_u loop, one problem per wavefront)",
                         "kernel_ms": step_ms,
                         "note": "algorithmic flops = problem-iterations x (T x 225 kflop backward + T x 7.3 kflop "
                                 "for the accepted rollout); fp64 MFMA peak = fp64 vector peak on MI355X"},
        }
        if cnt:
            line["roofline"]["counters"] = cnt
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, min(16, os.cpu_count() or 1))
            n_cpu = min(B, 2048)  # ~15 s on 16 cores
            from oracle import c_oracle as co  # the CPU baseline leg: the checker timed as a C port
            Mo = co.synthetic_model(T=TL)
            t1 = time.perf_counter()
            co.solve(Mo, co.Opts(), x0_h[:n_cpu], us_h[:n_cpu], nthreads=threads, trace=False)
            dt = time.perf_counter() - t1
            line["cpu_baseline"] = dict(value=n_cpu / dt, unit="solves/s", cores=threads, kind="port",
                                        sample="first %d problems of rank 0, C port of the reference algorithm "
                                               "(gcc -O2 -ffp-contract=off, OpenMP over problems), %.1f s wall"
                                               % (n_cpu, dt))
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def cartpole_roofline(B, iters, kernel_ms, chunked):
    """The roofline object of the cart-pole solve.  Live: the kernel time (device events) and everything
    derived from the iteration counts of this run.  From profiles/solve_counters.json (rocprofv3 PMC passes
    over this same command): measured HBM traffic and the VALU-busy fractions."""
    prob_iters = int(iters.sum())
    dur_s = kernel_ms * 1e-3
    cnt = load_counters("cartpole", B)
    # the chunked solve drops finished problems between launches, so a sector's 8 problems leave together
    # with at most one chunk's delay: the whole-solve factor does not apply to it
    sector = 1.0 if chunked else sector_granular_factor(iters)
    fused_bytes = prob_iters * (FUSED_READ_BYTES + FUSED_WRITE_BYTES)
    flops = prob_iters * (FLOP_BACKWARD + FLOP_ROLLOUT)
    staged = {"algorithmic_bytes_per_launch": prob_iters * STAGED_BYTES_PER_PROBLEM_ITER,
              "GBps_equivalent": prob_iters * STAGED_BYTES_PER_PROBLEM_ITER / dur_s / 1e9,
              "frac_of_hbm_peak": prob_iters * STAGED_BYTES_PER_PROBLEM_ITER / dur_s / 1e9 / HBM_PEAK_GBS,
              "note": "problem-iterations x %d B: what a derivative-kernel / backward / line-search pipeline that "
                      "streams the derivative tensors through HBM would move; the kernel does not move these "
                      "bytes (derivatives are evaluated in registers), so this is a comparison with that "
                      "design, not a utilisation" % STAGED_BYTES_PER_PROBLEM_ITER}
    fused = {"bytes_per_problem_iteration": FUSED_READ_BYTES + FUSED_WRITE_BYTES,
             "bytes_per_launch": fused_bytes,
             "sector_granularity_factor": None if chunked else sector,
             "expected_hbm_bytes_per_launch": fused_bytes * sector,
             "note": "xs, us in / k, K out (backward sweep) + xs, us, k, K in / xs, us out (accepted rollout); "
                     "memory moves 64-byte sectors = 8 neighbouring problems, alive until the slowest of the 8 "
                     "has ended, hence the factor"}
    traffic = cnt.get("hbm_bytes_per_launch") if cnt else None
    kernel = ("k_solve (whole update_u loop, one launch per step; its pack/unpack transposes are inside the "
              "timed span)" if not chunked else
              "k_solve in launches of 32 iterations, finished problems dropped in between (index-list "
              "conversions inside the timed span)")
    hbm_bound = B >= 65536  # one 64-problem wavefront on every SIMD: the ping-pong workspace streams through HBM
    if hbm_bound:
        moved = traffic if traffic else fused_bytes * sector
        roof = {"bound": "hbm", "achieved": moved / dur_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": moved / dur_s / 1e9 / HBM_PEAK_GBS,
                "basis": "PMC-measured bytes" if traffic else "fused-design bytes of this run's iteration counts"}
    else:
        # Up to 32768 problems every wavefront has a SIMD to itself and the launch lasts as long as its slowest
        # group of problems: max_iterations x T time steps, each a backward step and a rollout step whose inputs
        # are the outputs of the step before.  The bound of such a launch is the length of those loop-carried
        # chains of dependent instructions x the issue interval of dependent instructions; frac = that bound /
        # the measured time says how far the kernel is from the limit its data dependencies set.
        cp = load_critical_path()
        roof = {"bound": "dependent-instruction chain (latency)", "unit": "time steps/s of the slowest problem group"}
        if cp is not None:
            chain = cp["backward_step_chain"] + cp["rollout_step_chain"]
            step_s = chain * cp["dependent_issue_cycles"] / cp["shader_clock_hz"]
            n_steps = int(iters.max()) * T
            roof.update(achieved=n_steps / dur_s, peak=1.0 / step_s, frac=n_steps * step_s / dur_s,
                        critical_path={"ms": n_steps * step_s * 1e3, "time_steps": n_steps,
                                       "chain_instructions_per_time_step": chain,
                                       "backward_step_chain": cp["backward_step_chain"],
                                       "rollout_step_chain": cp["rollout_step_chain"],
                                       "dependent_issue_cycles": cp["dependent_issue_cycles"],
                                       "shader_clock_hz": cp["shader_clock_hz"], "probes": cp["probes"],
                                       "basis": "profiles/critical_path.json (scripts/chain_depth.py on "
                                                "scripts/micro/chain_probes.hip)"})
        else:
            roof.update(achieved=None, peak=None, frac=None)
        occ = {"note": "share of the chip's SIMD cycles with a vector instruction in flight (an occupancy, not a "
                       "distance from a limit: a faster kernel shows a lower value)"}
        if cnt and cnt.get("valu_busy_frac_of_chip") is not None:
            occ.update(frac_of_chip=cnt["valu_busy_frac_of_chip"],
                       valu_busy_frac_per_wave=cnt.get("valu_busy_frac_per_wave"),
                       simd_occupancy=cnt.get("simd_occupancy"),
                       lane_weighted_frac=cnt.get("valu_lane_weighted_frac_of_chip"),
                       basis="SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x waves / %d SIMDs (%s)"
                             % (N_SIMD, ", ".join(cnt.get("source", []))))
        roof["valu_issue"] = occ
    roof.update(traffic=traffic, kernel=kernel, kernel_ms=kernel_ms,
                hbm={"measured_bytes_per_launch": traffic,
                     "measured_GBps": traffic / dur_s / 1e9 if traffic else None,
                     "measured_frac_of_peak": traffic / dur_s / 1e9 / HBM_PEAK_GBS if traffic else None,
                     "peak_GBps": HBM_PEAK_GBS},
                fused_design=fused, staged_design=staged,
                fp64_flops={"achieved_tflops": flops / dur_s / 1e12, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                            "frac": flops / dur_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                            "flop_model": "problem-iterations x (90 kflop backward + 15 kflop per rollout)"})
    if cnt:
        # the PMC passes are a stored profile of one build: say which, and whether this run still matches it
        prof_ms = cnt.get("kernel_avg_ms")
        stale = bool(prof_ms) and abs(kernel_ms - prof_ms) > 0.05 * prof_ms
        roof["profile"] = {"kernel_avg_ms": prof_ms, "git_head": cnt.get("git_head"),
                           "stale": stale,
                           "note": "traffic, hbm.measured_* and valu_issue come from this stored rocprofv3 profile; "
                                   "stale = its kernel time differs from this run's by more than 5 %"}
        roof["counters"] = cnt
    return roof


class _Device(object):
    """What the timed loop needs from the device: the real thing on a GPU; with ILQR_BENCH_STUB=1 (a test hook,
    see stub_solve) host tensors and the wall clock, so that the launcher, the process group, the sharding and
    the gather run on a machine without a GPU."""

    def __init__(self, torch, local_rank, stub):
        self.torch, self.stub = torch, stub
        if stub:
            self.dev = torch.device("cpu")
        else:
            torch.cuda.set_device(local_rank)
            self.dev = torch.device("cuda", local_rank)

    def synchronize(self):
        if not self.stub:
            self.torch.cuda.synchronize()

    def mark(self):
        """A point in the current stream's time (device event) or, for the stub, in wall-clock time."""
        if self.stub:
            return time.perf_counter()
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def ms_between(self, a, b):
        return (b - a) * 1e3 if self.stub else a.elapsed_time(b)


def stub_solve(x0, us0, out):
    """ILQR_BENCH_STUB=1 only (CPU test of the N > 1 wiring): stands in for the device solve with a
    deterministic function of the inputs, so that the gathered slices can be told apart and checked."""
    B = x0.shape[0]
    out["J"][:B] = (x0 * x0).sum(1) + (us0 * us0).sum((1, 2))
    out["iters"][:B] = 1 + (x0[:, 0].abs() * 7).to(out["iters"].dtype)
    out["status"][:B] = 1
    out["xs"][:B] = x0[:, None, :]
    out["us"][:B] = us0


def bits_checksum(t):
    """Order-independent checksum of a float64 tensor's bits (wrapping int64 sum)."""
    import torch
    return t.contiguous().view(torch.int64).sum().reshape(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("cartpole", "large"), default="cartpole",
                    help="cartpole: the benchmark configuration (default); large: synthetic n=32, m=8, T=200")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None,
                    help="problems per GPU (default: %d cart-pole problems, 8192 large-state problems)" % B_PER_GPU)
    ap.add_argument("--total", type=int, default=None,
                    help="strong scaling: this many problems in all, cut into contiguous slices over the ranks "
                         "(1048576 = the million-problem configuration)")
    ap.add_argument("--cpu-sample", type=int, default=B_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher around us: be the launcher
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if os.environ.get("ILQR_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ILQR_BENCH_WATCHDOG"]), exit=True)
    if args.workload == "large":
        return main_large(args)

    import torch
    import torch.distributed as dist
    from ilqr_amd import compaction, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d runs under a launcher with WORLD_SIZE=%d" % (args.gpus, world))
    # ILQR_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks
    # (ranks share devices; RCCL refuses that).  The measured configuration is always nccl (= RCCL).
    backend = os.environ.get("ILQR_BENCH_BACKEND", "nccl")
    stub = os.environ.get("ILQR_BENCH_STUB") == "1"
    if stub and backend == "nccl":
        raise SystemExit("ILQR_BENCH_STUB=1 (no device solve) needs ILQR_BENCH_BACKEND=gloo")
    if backend != "nccl" and not stub:
        local_rank %= torch.cuda.device_count()
    D = _Device(torch, local_rank, stub)
    dev = D.dev
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.total is not None:  # strong scaling: one global batch (generated per 16384-problem block, so that a
        # rank only materialises its own slice), cut into contiguous slices
        sizes = sharding.shard_sizes(args.total, world)
        lo, hi = sharding.shard_bounds(args.total, world, rank)
        blocks = range(lo // B_PER_GPU, (hi - 1) // B_PER_GPU + 1)
        parts = [make_batch(B_PER_GPU, blk) for blk in blocks]
        off = lo - blocks[0] * B_PER_GPU
        x0_h = np.concatenate([p[0] for p in parts])[off:off + hi - lo]
        us_h = np.concatenate([p[1] for p in parts])[off:off + hi - lo]
        B = hi - lo
    else:
        B = args.batch if args.batch is not None else B_PER_GPU
        sizes = [B] * world
        x0_h, us_h = make_batch(B, rank)
    x0 = torch.tensor(x0_h, device=dev)
    us0 = torch.tensor(us_h, device=dev)
    chunked = B >= compaction.COMPACT_MIN_BATCH  # the product's policy for batches that fill the GPU
    ws = [None]
    # Result tensors: with more than one rank two sets alternate, so that the gather of one solve's
    # results (on the collective's stream) overlaps the next solve.  Sets hold max(sizes) rows (slices of
    # unequal size are gathered padded).
    n_sets = 2 if world > 1 else 1
    rows = max(sizes)
    f64, i32 = torch.float64, torch.int32
    out_sets = [dict(xs=torch.zeros((rows, T + 1, 5), dtype=f64, device=dev),
                     us=torch.zeros((rows, T, 1), dtype=f64, device=dev),
                     J=torch.zeros(rows, dtype=f64, device=dev), mu=torch.zeros(rows, dtype=f64, device=dev),
                     iters=torch.zeros(rows, dtype=i32, device=dev),
                     status=torch.zeros(rows, dtype=i32, device=dev)) for _ in range(n_sets)]
    if stub:
        def solve_into(out):
            stub_solve(x0, us0, out)
    else:
        from ilqr_amd import _hip
        Q, R, Qt, xg = cartpole_cost()
        M = _hip.Model(T, Q, R, Qt, xg, None, [-5.0], [5.0], dt=0.05)
        O = _hip.Opts()
        delta = torch.empty(B, dtype=f64, device=dev)
        k = torch.empty((B, T, 1), dtype=f64, device=dev)
        K = torch.empty((B, T, 1, 5), dtype=f64, device=dev)

        def solve_into(out):
            # a freshly constructed solver per step: delta_ = 2, k_ = K_ = 0 (ilqr_core.py:58-65)
            delta.fill_(2.0)
            k.zero_()
            K.zero_()
            mine = {name: t[:B] for name, t in out.items()}  # (a view of the leading rows is contiguous)
            if chunked:
                res = _hip.solve_chunked(M, O, x0, us0, compaction.COMPACT_CHUNK, delta=delta, k=k, K=K,
                                         workspace=ws[0], out=mine)
            else:
                res = _hip.solve(M, O, x0, us0, delta=delta, k=k, K=K, workspace=ws[0], out=mine)
            ws[0] = res["workspace"]

    pipe = sharding.GatherPipeline(depth=n_sets, dst=0) if world > 1 else None
    # rehearsal only: gloo's gathers of GPU tensors hang behind a running kernel
    job = ShardedSolveJob(solve_into, out_sets, pipe,
                          before_submit=D.synchronize if backend != "nccl" else None,
                          mark=D.mark if world > 1 else None)

    def barrier():
        D.synchronize()
        if world > 1:
            dist.barrier()
        D.synchronize()

    ev = [[None, None] for _ in range(args.steps)]

    def record(i, what):
        ev[i][0 if what == "start" else 1] = D.mark()
    elapsed = run_steps(job, args.steps, args.warmup, barrier, record=record)
    # from the end of the last solve to the arrival of its results on rank 0: the gather no solve overlaps
    gather_exposed_ms = D.ms_between(job.solve_end, job.drained) if world > 1 else None
    step_ms_events = float(np.mean([D.ms_between(a, b) for a, b in ev]))

    t = torch.tensor([elapsed], dtype=f64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    out_buf = out_sets[job.last_slot()]
    # every rank's checksum of the costs it produced in the last step, collected on rank 0 beside the data
    sums = None
    if world > 1:
        mine = bits_checksum(out_buf["J"][:B])
        sums = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, sums, dst=0)
    iters = out_buf["iters"][:B].cpu().numpy().astype(np.int64)
    status = out_buf["status"][:B].cpu().numpy()
    J = out_buf["J"][:B].cpu().numpy()
    if rank == 0:
        total = int(sum(sizes))
        ms_per_step = elapsed / args.steps * 1e3
        solves_per_s = total * args.steps / elapsed
        max_iters = int(iters.max())
        gathered_ok = None
        if pipe is not None:
            # the last step's results as rank 0 received them: the slice of EVERY rank must carry the checksum
            # that rank computed before sending, and rank 0's own slice must be the tensor it solved into
            g = pipe.result(job.last_slot(), "J").reshape(world, rows)
            per_rank = [bool(torch.equal(bits_checksum(g[r, :sizes[r]]), sums[r])) for r in range(world)]
            gathered_ok = bool(all(per_rank) and torch.equal(g[0, :B], out_buf["J"][:B]))
        if args.total is not None:
            workload = ("%d cart-pole swing-up problems in all, cut into %d contiguous slices (%s per GPU), T=100, "
                        "n=5, m=1, control limits +-5 active, <=200 iterations, tol 1e-4, 30-alpha schedule"
                        % (total, world, "/".join(str(s) for s in sorted(set(sizes)))))
        else:
            workload = ("batched cartpole swing-up, %d random initial states per GPU, T=100, n=5, m=1, "
                        "control limits +-5 active, <=200 iterations, tol 1e-4, 30-alpha schedule" % B)
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
            "scaling": "strong" if args.total is not None else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "problems_per_gpu": B, "problems_total": total, "horizon": T,
                       "parallelism": "batch-sharded x%d, results gathered on rank 0" % world,
                       "solve": "chunked (launches of %d iterations)" % compaction.COMPACT_CHUNK if chunked
                                else "single launch",
                       "mean_iterations_per_solve": float(iters.mean()), "max_iterations": max_iters,
                       "converged_fraction": float((status == 1).mean()), "mean_final_cost": float(J.mean())},
        }
        if not stub:
            line["roofline"] = cartpole_roofline(B, iters, step_ms_events, chunked)
        else:
            line["config"]["stub"] = "ILQR_BENCH_STUB=1: no device solve ran; this line is not a measurement"
        if world > 1:
            line["config"]["ranks_seen"] = int(dist.get_world_size())
            line["config"]["collective_backend"] = backend
            line["config"]["gather_check"] = gathered_ok
            line["config"]["gather_check_per_rank"] = per_rank
            line["config"]["gather_exposed_ms_last_step"] = gather_exposed_ms
        if world == 1 and not args.no_cpu_baseline and not stub:
            threads = max(1, min(16, os.cpu_count() or 1))
            line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, B), threads)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
