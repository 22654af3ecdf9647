"""One rank of the two-real-ranks test (tests/test_two_ranks_gpu.py).  NOT a test module: it is
started as a fresh interpreter per rank -- nothing re-execs a process that has initialised the
GPU -- the way bench.py starts its rank children.

Every rank opens its OWN bmc_ctx on the device it is told to use, plans its persistent launches
for ``--cu-limit`` CUs (two ranks sharing one MI355X take half the chip each, so that both
launches are planned to be resident side by side), runs its chain_block of the job with the REAL
sampler (bmc_gibbs_run_device into a torch tensor), copies the block to the host and pools
through pybmc_amd.chains.pool_samples over gloo.  Rank 0 saves the pooled array."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--n-chains", type=int, required=True)
    ap.add_argument("--iters", type=int, required=True)
    ap.add_argument("--n", type=int, required=True)
    ap.add_argument("--k", type=int, required=True)
    ap.add_argument("--cu-limit", type=int, default=0)
    ap.add_argument("--base-seed", type=int, default=11)
    ap.add_argument("--runs", type=int, default=1)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(a.port)
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    try:
        from pybmc_amd import _lib
        from pybmc_amd.chains import chain_block, chain_seeds, pool_samples
        from pybmc_amd.synthetic import synth_problem

        p = synth_problem(a.n, a.k + 1, a.k, seed=0)
        ctx = _lib.Context(a.device)
        if a.cu_limit:
            ctx.set_tuning(cu_limit=a.cu_limit)
        ctx.set_problem(p["y"], p["X"])
        ctx.set_prior(*p["prior"])
        mine = chain_block(a.n_chains, a.world, a.rank)
        dev = torch.device("cuda", a.device)
        out = torch.empty((len(mine), a.iters, a.k + 1), dtype=torch.float64, device=dev)
        dist.barrier()           # both ranks launch at about the same time: real co-residency
        stats = None
        first = None
        unstable = 0             # runs whose output differs from this rank's first run
        for r in range(a.runs):
            if mine:
                stats = ctx.gibbs_run_device(len(mine), a.iters, chain_seeds(a.base_seed, mine),
                                             out.data_ptr())
                if a.runs > 1:
                    cur = out.cpu()
                    if first is None:
                        first = cur
                    elif not torch.equal(cur, first):
                        unstable += 1
        host = out.cpu()         # gloo pools host tensors
        pooled = pool_samples(host, a.n_chains)
        info = [None] * a.world
        dist.all_gather_object(info, {"rank": a.rank, "chains": mine,
                                      "groups": stats["groups_per_chain"] if stats else 0,
                                      "waves": stats["waves_per_group"] if stats else 0,
                                      "launches": stats["launches"] if stats else 0,
                                      "loop_ms": stats["loop_ms"] if stats else 0.0,
                                      "xcd_local": stats["xcd_local_chains"] if stats else 0,
                                      "unstable_runs": unstable})
        if a.rank == 0:
            np.save(a.out, pooled.numpy())
            import json
            with open(a.out + ".json", "w") as f:
                json.dump(info, f)
        ctx.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
