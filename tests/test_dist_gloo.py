"""CPU, world_size=2 over gloo: the N>1 plumbing (channel sharding, PCM fan-out, soft-bit gather)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jaero_amd import dist as jd


def test_shard_ranges_cover_and_are_disjoint():
    for n in (1, 7, 64, 4096, 32768, 100):
        for w in (1, 2, 3, 8):
            r = [jd.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [h - l for l, h in r]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, nch, nsamp, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = jd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    frames = None
    if rank == 0:
        frames = torch.arange(nsamp * nch, dtype=torch.int32).reshape(nsamp, nch).to(torch.int16)
    mine = jd.fan_out_pcm(frames, nch, nsamp, src=0, device=torch.device("cpu"))
    lo, hi = jd.shard_range(nch, rank, world)
    want = torch.arange(nsamp * nch, dtype=torch.int32).reshape(nsamp, nch).to(torch.int16)[:, lo:hi]
    ok = bool(torch.equal(mine, want))
    # each rank "demodulates" its slice: soft slot k of channel c = (c*7 + k) % 256, count = c % 5
    cap = 6
    ch = torch.arange(lo, hi)
    soft = ((ch[:, None] * 7 + torch.arange(cap)[None, :]) % 256).to(torch.int16)
    counts = (ch % 5).to(torch.int32)
    sa, ca = jd.gather_softbits(soft, counts, nch, dst=0)
    if rank == 0:
        allc = torch.arange(nch)
        ok &= bool(torch.equal(sa, ((allc[:, None] * 7 + torch.arange(cap)[None, :]) % 256).to(torch.int16)))
        ok &= bool(torch.equal(ca, (allc % 5).to(torch.int32)))
    else:
        ok &= sa is None and ca is None
    dist.barrier()
    q.put((rank, ok))
    dist.destroy_process_group()


def test_fan_out_and_gather_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 11, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _bench_worker(rank, world, port, q):
    """bench.py's own rank plumbing (run_timed / rank_fields) over gloo: barrier on both sides of exactly K steps, every rank's wall time
    gathered, the slowest one is the job's."""
    import argparse
    import time

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench

    bench.ARGS = argparse.Namespace(gpus=world)
    r, w, _ = jd.init_from_env("gloo")
    calls = []

    def step(i):
        calls.append(i)
        time.sleep(0.01 * (rank + 1))  # rank 1 is the slow one

    dt, dts = bench.run_timed(step, 2, 5, world, torch.device("cpu"))
    f = bench.rank_fields(world, False, dts, 1000.0, scale=1.0)
    ok = calls == list(range(7)) and len(dts) == world and dt >= max(dts) and dts[1] > dts[0] and f["ranks"] == world
    ok = ok and f["per_rank_rate"][0] > f["per_rank_rate"][1]
    dist.barrier()
    q.put((rank, ok))
    dist.destroy_process_group()


def test_bench_rank_plumbing_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no launcher around it re-executes itself under torch.distributed.run with N ranks on 127.0.0.1; under a
    launcher (WORLD_SIZE set) or at N = 1 it does not; a WORLD_SIZE that contradicts --gpus is refused (no silent single-rank run)."""
    import argparse
    import subprocess
    import sys

    import bench

    seen = {}

    def fake_call(cmd):
        seen["cmd"] = cmd
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.ARGS = argparse.Namespace(gpus=4)
    with pytest.raises(SystemExit) as e:
        bench.maybe_spawn()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.maybe_spawn()
    assert not seen
    monkeypatch.delenv("WORLD_SIZE")
    bench.ARGS = argparse.Namespace(gpus=1)
    bench.maybe_spawn()
    assert not seen
