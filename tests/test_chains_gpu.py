"""Single-rank run of the multi-GPU driver (-m gpu): torch owns the output tensor, the C ABI
writes into it (bmc_gibbs_run_device), chains are seeded by global id."""
import numpy as np
import pytest

from gpu_common import gpu_ctx
from pybmc_amd.chains import chain_seeds, posterior_summary, run_chains
from pybmc_amd.synthetic import synth_problem

pytestmark = pytest.mark.gpu


def test_run_chains_single_rank_matches_host_path():
    import torch
    ctx = gpu_ctx()
    p = synth_problem(5000, 9, 8, seed=6)
    ctx.set_problem(p["y"], p["X"])
    ctx.set_prior(*p["prior"])
    T, C = 3000, 4
    pooled, stats = run_chains(ctx, C, T, base_seed=9)
    assert isinstance(pooled, torch.Tensor) and pooled.is_cuda and pooled.shape == (C, T, 9)
    host, _ = ctx.gibbs_run(C, T, seeds=chain_seeds(9, list(range(C))))
    assert np.array_equal(pooled.cpu().numpy(), host)       # device-output path == host-output path
    Vt_hat = p["Vt"] / p["S_hat"][:, None]
    s = posterior_summary(host, Vt_hat, burn=500)
    assert abs(s["weights_mean"].sum() - 1) < 1e-9 and abs(s["sigma_mean"] - 0.1) < 0.01
    assert stats["n_chains"] == C


def test_one_hip_runtime_in_the_process():
    """torch and libpybmc_amd must share one libamdhip64 (device pointers cross the ABI)."""
    import torch  # noqa: F401
    ctx = gpu_ctx()  # noqa: F841
    with open("/proc/self/maps") as f:
        libs = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert len(libs) == 1, libs


def test_bench_rccl_leg_with_one_rank():
    """The RCCL leg of bench.py (init, barrier, all_gather_into_tensor on the sampler's output
    block, all_reduce of the time, all_gather_object, destroy) executed with backend "nccl" on
    this box's one GPU: a world of one rank (--force-dist).  More ranks need more GPUs."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist",
                        "--steps", "2", "--warmup", "1", "--iters", "3000", "--no-extra",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["rccl"]["backend"] == "nccl" and line["rccl"]["world"] == 1
    assert line["rccl"]["allgather_ms"] > 0 and line["rccl"]["allgather_bytes_per_rank"] == 3000 * 33 * 8
    assert line["value"] > 1e5 and line["n_gpus"] == 1 and line["roofline"]["bound"] == "latency"
