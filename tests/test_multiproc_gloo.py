"""The N > 1 path on CPU: world_size 2 (and a ragged 3-chain case) over gloo.

The chain -> rank partition, the per-chain seeds and the single all-gather that
pools the per-rank sample blocks are exercised exactly as bench.py / run_chains use
them on RCCL; only the per-rank sampler is replaced by a deterministic stand-in
(there is no GPU here), so the test checks that the pooled tensor is independent
of the number of ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pybmc_amd.chains import chain_block, chain_seeds, pool_samples

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
