"""The N > 1 path on CPU: two processes over the gloo backend shard a batch, each
solves its slice (with the CPU oracle standing in for the device solve), and the
gathered result must equal the unsharded solve bit for bit."""
import os
import socket

import numpy as np
import pytest

from ilqr_amd import sharding


def test_shard_bounds_cover_the_batch():
    for total in (0, 1, 7, 64, 16384, 1000003):
        for world in (1, 2, 3, 8):
            bounds = [sharding.shard_bounds(total, world, r) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == total
            assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
            sizes = sharding.shard_sizes(total, world)
            assert sum(sizes) == total and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, T, out_dir):
    import torch
    import torch.distributed as dist
    from oracle import c_oracle as co
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x0, us0 = co.cartpole_batch(B, T=T, seed_x=71, seed_u=72)
        Mo = co.cartpole_model(T=T)

        def local_solve(x0_s, us_s):
            r = co.solve(Mo, co.Opts(max_iter=15), x0_s, us_s, trace=False)
            return {k: torch.tensor(r[k]) for k in ("xs", "us", "J", "iters", "status")}
        res, gathered = sharding.solve_sharded(local_solve, x0, us0, dst=0)
        lo, hi = sharding.shard_bounds(B, world, rank)
        assert res["J"].shape[0] == hi - lo
        if rank == 0:
            np.savez(os.path.join(out_dir, "gathered.npz"), **{k: v.numpy() for k, v in gathered.items()})
        else:
            assert gathered is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [37, 64])
def test_two_rank_gloo_solve_equals_unsharded(tmp_path, B):
    import torch.multiprocessing as mp
    from oracle import c_oracle as co
    T = 20
    mp.spawn(_worker, args=(2, _free_port(), B, T, str(tmp_path)), nprocs=2, join=True)
    g = np.load(tmp_path / "gathered.npz")
    x0, us0 = co.cartpole_batch(B, T=T, seed_x=71, seed_u=72)
    ref = co.solve(co.cartpole_model(T=T), co.Opts(max_iter=15), x0, us0, trace=False)
    for k in ("xs", "us", "J", "iters", "status"):
        assert g[k].shape == ref[k].shape and np.array_equal(g[k], ref[k]), k


def _pipeline_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = sharding.GatherPipeline(depth=2, dst=0)
        sets = [dict(J=torch.empty(5, dtype=torch.float64), it=torch.empty((5, 3), dtype=torch.int32))
                for _ in range(2)]
        seen = []
        for step in range(5):
            slot = step % 2
            pipe.reclaim(slot)
            if rank == 0 and step >= 2:  # the set's previous contents have arrived and are still intact
                seen.append((step - 2, pipe.result(slot, "J").clone(), pipe.result(slot, "it").clone()))
            sets[slot]["J"].copy_(torch.arange(5, dtype=torch.float64) + 100 * step + 10 * rank)
            sets[slot]["it"].fill_(1000 * step + rank)
            pipe.submit(slot, sets[slot])
        pipe.drain()
        if rank == 0:
            for step in (3, 4):
                seen.append((step, pipe.result(step % 2, "J").clone(), pipe.result(step % 2, "it").clone()))
            for step, J, it in seen:
                want_J = torch.cat([torch.arange(5, dtype=torch.float64) + 100 * step + 10 * r for r in range(world)])
                want_it = torch.cat([torch.full((5, 3), 1000 * step + r, dtype=torch.int32) for r in range(world)])
                assert torch.equal(J, want_J) and torch.equal(it, want_it), step
            assert sorted(s for s, _, _ in seen) == [0, 1, 2, 3, 4]
            open(os.path.join(out_dir, "ok"), "w").close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_pipelined_gathers_deliver_every_set(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _bench_job_worker(rank, world, port, total, T, out_dir):
    """bench.py's own step wiring (ShardedSolveJob + run_steps + GatherPipeline) on two gloo ranks, with the
    device solve replaced by the oracle on CPU tensors: slices of unequal size, result sets that alternate."""
    import sys
    import torch
    import torch.distributed as dist
    from oracle import c_oracle as co
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x0, us0 = co.cartpole_batch(total, T=T, seed_x=81, seed_u=82)
        Mo = co.cartpole_model(T=T)
        sizes = sharding.shard_sizes(total, world)
        lo, hi = sharding.shard_bounds(total, world, rank)
        B, rows = hi - lo, max(sizes)
        out_sets = [dict(xs=torch.zeros((rows, T + 1, 5), dtype=torch.float64), us=torch.zeros((rows, T, 1), dtype=torch.float64),
                         J=torch.zeros(rows, dtype=torch.float64), iters=torch.zeros(rows, dtype=torch.int32),
                         status=torch.zeros(rows, dtype=torch.int32)) for _ in range(2)]
        calls = []

        def solve_into(out):
            # every step solves a different problem family (more iterations allowed), so that a set that was
            # not rewritten, or a gather of the wrong set, shows in the results
            n_it = 3 + len(calls)
            r = co.solve(Mo, co.Opts(max_iter=n_it), x0[lo:hi], us0[lo:hi], trace=False)
            for key in ("xs", "us", "J", "iters", "status"):
                out[key][:B].copy_(torch.tensor(r[key]))
            calls.append(id(out))
        pipe = sharding.GatherPipeline(depth=2, dst=0)
        job = bench.ShardedSolveJob(solve_into, out_sets, pipe)
        elapsed = bench.run_steps(job, steps=3, warmup=1, barrier=dist.barrier)
        assert elapsed > 0 and job.n_steps == 4
        assert calls == [id(out_sets[i % 2]) for i in range(4)]  # the two sets really alternate
        if rank == 0:
            for step in (2, 3):  # the last two steps are still held by the two sets
                ref = co.solve(Mo, co.Opts(max_iter=3 + step), x0, us0, trace=False)
                for key in ("xs", "us", "J", "iters", "status"):
                    g = pipe.result(step % 2, key)
                    g = g.reshape((world, rows) + tuple(g.shape[1:]))
                    got = torch.cat([g[r, :sizes[r]] for r in range(world)]).numpy()
                    assert got.shape == ref[key].shape and np.array_equal(got, ref[key]), (step, key)
            open(os.path.join(out_dir, "ok"), "w").close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_bench_step_wiring(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_bench_job_worker, args=(2, _free_port(), 37, 20, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_sector_granular_factor():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert bench.sector_granular_factor([5] * 16) == 1.0
    assert bench.sector_granular_factor([10] + [0] * 7) == 8.0
    assert abs(bench.sector_granular_factor([4, 2] * 4 + [1] * 8) - (32 + 8) / 32.0) < 1e-12


def _run_bench(argv, env_extra, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr.decode()


@pytest.mark.parametrize("argv, sizes", [(["--gpus", "2", "--batch", "48"], [48, 48]),
                                         (["--gpus", "3", "--total", "100"], [34, 33, 33])])
def test_bench_launches_its_own_ranks_and_checks_every_slice(argv, sizes):
    """`bench.py --gpus N` without a launcher: the parent starts N ranks, relays ONE JSON line, and the line
    says how many ranks the collective saw and that every rank's gathered slice carries that rank's checksum.
    (ILQR_BENCH_STUB=1 replaces the device solve, gloo the RCCL backend: no GPU here.)"""
    rc, lines, err = _run_bench(argv + ["--steps", "2", "--warmup", "1"],
                                {"ILQR_BENCH_BACKEND": "gloo", "ILQR_BENCH_STUB": "1"})
    assert rc == 0, err
    assert len(lines) == 1
    line = lines[0]
    cfg = line["config"]
    assert line["n_gpus"] == len(sizes) and cfg["ranks_seen"] == len(sizes)
    assert cfg["problems_total"] == sum(sizes) and cfg["problems_per_gpu"] == sizes[0]
    assert cfg["gather_check"] is True and cfg["gather_check_per_rank"] == [True] * len(sizes)
    assert cfg["gather_exposed_ms_last_step"] >= 0.0
    assert line["value"] > 0 and line["steps"] == 2 and line["warmup"] == 1


def test_bench_launcher_reports_a_failed_rank():
    """A rank that dies makes the parent exit non-zero and print no result line (here: every rank refuses the
    stub without the gloo backend)."""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"ILQR_BENCH_STUB": "1"},
                                timeout=120)
    assert rc != 0 and lines == []
    assert "exited with status" in err
