"""Two REAL ranks on the one GPU of the box (-m gpu): SURVEY.md 8(e)'s claim -- the pooled
posterior does not depend on the number of ranks -- with the HIP sampler in every rank instead
of the stand-ins of tests/test_multiproc_gloo.py.

Two fresh interpreters (tests/rank_child.py), backend gloo, both on device 0, each with its own
bmc_ctx planning for half the chip (bmc_tuning.cu_limit = 128) so that the two persistent
launches are co-resident; each runs its chain_block with bmc_gibbs_run_device, copies its block
to the host and pools through pool_samples.  The pooled tensor must be BIT-IDENTICAL to one
process running all chains with the same global seeds.  The reference has no counterpart (one
chain per train(), pybmc/bmc.py:188-193)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from gpu_common import gpu_ctx
from pybmc_amd.chains import chain_block, chain_seeds
from pybmc_amd.synthetic import synth_problem

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(tmp_path, world, n_chains, iters, n, k, cu_limit, runs=1, limit_by_env=False):
    port = free_port()
    out = str(tmp_path / f"pooled_{n}_{n_chains}.npy")
    env = {kk: v for kk, v in os.environ.items()
           if kk not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if limit_by_env:     # each rank's share of the GPU through PYBMC_AMD_CU_LIMIT (read by bmc_create)
        env["PYBMC_AMD_CU_LIMIT"] = str(cu_limit)
        cu_limit = 0
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_child.py"),
                               "--rank", str(r), "--world", str(world), "--port", str(port),
                               "--n-chains", str(n_chains), "--iters", str(iters), "--n", str(n),
                               "--k", str(k), "--cu-limit", str(cu_limit), "--runs", str(runs),
                               "--out", out],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=420)
            outs.append(o.decode("utf-8", "replace"))
    finally:
        for p in procs:          # exactly the children started above
            if p.poll() is None:
                p.kill()
                p.wait()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    with open(out + ".json") as f:
        info = json.load(f)
    return np.load(out), info


def describe_difference(a, b):
    """Which chains differ, from which iteration on and by how much (assertion message)."""
    out = []
    for c in range(a.shape[0]):
        d = np.abs(a[c] - b[c]).max(axis=1)
        nz = np.nonzero(d)[0]
        out.append(f"chain {c}: " + ("equal" if len(nz) == 0 else
                                     f"{len(nz)} rows differ, first t={nz[0]} |d|={d[nz[0]]:.3e}, max {d.max():.3e}"))
    return "; ".join(out)


def single_process(n_chains, iters, n, k):
    ctx = gpu_ctx()
    p = synth_problem(n, k + 1, k, seed=0)
    ctx.set_problem(p["y"], p["X"])
    ctx.set_prior(*p["prior"])
    ref, st = ctx.gibbs_run(n_chains, iters, seeds=chain_seeds(11, list(range(n_chains))))
    return ref, st


@pytest.mark.parametrize("n_chains", [4, 3])
def test_two_ranks_single_workgroup_chains(tmp_path, n_chains):
    """N = 629 (the notebook's size): every chain is one workgroup, no co-residency needed."""
    iters, n, k = 2000, 629, 3
    pooled, info = run_ranks(tmp_path, 2, n_chains, iters, n, k, cu_limit=128)
    ref, st = single_process(n_chains, iters, n, k)
    assert pooled.shape == ref.shape == (n_chains, iters, k + 1)
    assert np.array_equal(pooled, ref), describe_difference(pooled, ref)
    assert [i["chains"] for i in info] == [chain_block(n_chains, 2, r) for r in range(2)]


@pytest.mark.parametrize("n_chains", [4, 3])
def test_two_ranks_at_the_headline_size(tmp_path, n_chains):
    """C2 (N = 10 000, K = 32): 32 workgroups x 5 waves per chain exchanging partial sums every
    iteration, two (or 2 + 1) chains per rank, the two ranks' persistent launches side by side
    on the two halves of the chip.  Several runs per rank so that the launches overlap in time."""
    iters, n, k = 20000, 10000, 32
    pooled, info = run_ranks(tmp_path, 2, n_chains, iters, n, k, cu_limit=128, runs=3,
                             limit_by_env=(n_chains == 3))
    ref, st = single_process(n_chains, iters, n, k)
    assert st["groups_per_chain"] == 32
    # (half the chip per rank = 4 slots: a chain's 32 groups then normally span two XCDs and
    # exchange at agent scope, while the single process is XCD-local -- the bits are the same)
    assert all(i["groups"] == 32 and i["waves"] == st["waves_per_group"] for i in info), info
    assert pooled.shape == ref.shape == (n_chains, iters, k + 1)
    if not np.array_equal(pooled, ref):
        # say which side moved: the single-process run repeated, and on a fresh context
        ref2, _ = single_process(n_chains, iters, n, k)
        from pybmc_amd import _lib
        fresh = _lib.Context(0)
        p = synth_problem(n, k + 1, k, seed=0)
        fresh.set_problem(p["y"], p["X"])
        fresh.set_prior(*p["prior"])
        ref3, _ = fresh.gibbs_run(n_chains, iters, seeds=chain_seeds(11, list(range(n_chains))))
        fresh.close()
        pytest.fail("pooled != single process: " + describe_difference(pooled, ref)
                    + " | single process repeated: " + describe_difference(ref2, ref)
                    + " | fresh context: " + describe_difference(ref3, ref)
                    + " | pooled vs fresh: " + describe_difference(pooled, ref3))
    # and the pooled posterior is the posterior: sigma on the generating 0.1
    assert abs(pooled[:, iters // 5:, -1].mean() - 0.1) < 0.01
