"""The N > 1 path on CPU: world_size 2 (and a ragged 3-chain case) over gloo.

The chain -> rank partition, the per-chain seeds and the single all-gather that
pools the per-rank sample blocks are exercised exactly as bench.py / run_chains use
them on RCCL; only the per-rank sampler is replaced by a deterministic stand-in
(there is no GPU here), so the test checks that the pooled tensor is independent
of the number of ranks.  ``run_chains`` itself is driven through a stand-in context
(same ``gibbs_run_device(n, iters, seeds, out_ptr)`` call as pybmc_amd._lib.Context),
including the reuse of one output buffer over several runs and ragged chain blocks, and
``python bench.py --gpus 2`` is started plainly to check that it spawns its own ranks."""
import ctypes
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pybmc_amd.chains import chain_block, chain_seeds, pool_samples, run_chains

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

T, K1 = 16, 5


def fake_chain(seed):
    """Stand-in for one chain's [T, k+1] block: a pure function of the chain's seed."""
    g = np.random.Generator(np.random.PCG64(int(seed)))
    return torch.from_numpy(g.standard_normal((T, K1)))


def expected(n_chains):
    return torch.stack([fake_chain(s) for s in chain_seeds(5, list(range(n_chains)))])


def worker(rank, world, port, n_chains, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = chain_block(n_chains, world, rank)
        seeds = chain_seeds(5, mine)
        local = (torch.stack([fake_chain(s) for s in seeds]) if mine
                 else torch.empty((0, T, K1), dtype=torch.float64))
        pooled = pool_samples(local, n_chains)
        ok = pooled.shape == (n_chains, T, K1) and torch.equal(pooled, expected(n_chains))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_chains,world", [(8, 2), (3, 2), (2, 2)])
def test_pooling_is_independent_of_rank_count(n_chains, world):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, n_chains, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world))


def test_single_process_pool_is_identity():
    x = torch.arange(2 * T * K1, dtype=torch.float64).reshape(2, T, K1)
    assert pool_samples(x, 2) is x


class StandInCtx:
    """What run_chains needs of a context: k, device, gibbs_run_device writing this rank's
    [n, iters, k+1] block at a raw pointer (here: host memory of a CPU tensor)."""

    def __init__(self, k):
        self.k, self.device, self.torch_device = k, 0, torch.device("cpu")
        self.runs = 0

    def gibbs_run_device(self, n_chains, iters, seeds, out_ptr):
        assert iters == T
        n = n_chains * iters * (self.k + 1)
        buf = np.ctypeslib.as_array((ctypes.c_double * n).from_address(out_ptr))
        buf = buf.reshape(n_chains, iters, self.k + 1)
        for c, s in enumerate(seeds):
            buf[c] = fake_chain(s).numpy() + self.runs      # a different block every run
        self.runs += 1
        return {"n_chains": n_chains, "iterations": iters}


def run_chains_worker(rank, world, port, n_chains, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = StandInCtx(K1 - 1)
        mine = chain_block(n_chains, world, rank)
        out = torch.full((len(mine), T, K1), -7.0, dtype=torch.float64)
        ok = True
        for run in range(3):            # the SAME buffer handed to three runs in a row
            pooled, stats = run_chains(ctx, n_chains, T, base_seed=5, out=out)
            ok = ok and pooled.shape == (n_chains, T, K1)
            ok = ok and torch.equal(pooled, expected(n_chains) + run)
            ok = ok and ((stats is None) == (len(mine) == 0))
        pooled2, _ = run_chains(ctx, n_chains, T, base_seed=5)      # buffer allocated inside
        ok = ok and torch.equal(pooled2, expected(n_chains) + 3)
        try:
            run_chains(ctx, n_chains, T, base_seed=5, out=torch.empty((len(mine) + 1, T, K1),
                                                                     dtype=torch.float64))
            ok = False
        except ValueError:
            pass
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_chains,world", [(4, 2), (3, 2), (1, 2)])
def test_run_chains_through_a_stand_in_context(n_chains, world):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = free_port()
    procs = [ctx.Process(target=run_chains_worker, args=(r, world, port, n_chains, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world))


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts two rank children, they
    rendezvous on 127.0.0.1 (gloo here, with the sampler stood in: --dry-run-cpu), pool their
    chains, and the parent relays ONE JSON line that names the world and its devices."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--iters", "8", "--k", "3", "--dry-run-cpu",
                        "--rank-timeout-s", "240"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["dry_run"] is True and line["value"] is None and line["n_gpus"] == 2
    assert line["rccl"]["world"] == 2 and line["rccl"]["backend"] == "gloo"
    assert len(line["rccl"]["devices"]) == 2 and len(line["rccl"]["allgather_ms_per_rank"]) == 2
    assert line["rccl"]["allgather_bytes_per_rank"] == 1 * 8 * 4 * 8
    assert "roofline" not in line and "cpu_baseline" not in line     # nothing was measured


def test_bench_parent_fails_fast_when_a_rank_dies():
    """A rank that dies before the rendezvous (`--fail-rank 1`: what a missing device or an RCCL
    init error looks like) must not leave rank 0 waiting in init_process_group until torch's
    timeout: the parent polls every child, stops the others at once, names the failed rank with
    the tail of its output, prints no JSON line and exits non-zero."""
    import time
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--iters", "8", "--k", "3", "--dry-run-cpu",
                        "--fail-rank", "1", "--rank-timeout-s", "240"],
                       env=env, capture_output=True, text=True, timeout=300)
    took = time.monotonic() - t0
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert took < 120, f"the parent waited {took:.0f} s for a rank that was already dead"
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "rank 1 exited with status 3" in r.stderr and "--fail-rank test hook" in r.stderr
