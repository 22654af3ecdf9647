"""Diagnostic: where do two-rank chains at C2 differ from the single-process run?"""
import os, sys, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pybmc_amd import _lib
from pybmc_amd.chains import chain_seeds
from pybmc_amd.synthetic import synth_problem
import test_two_ranks_gpu as T2

def first_diff(a, b, tag):
    for c in range(a.shape[0]):
        d = np.abs(a[c] - b[c]).max(axis=1)
        nz = np.nonzero(d)[0]
        print(tag, "chain", c, "equal" if len(nz) == 0 else
              f"first diff at t={nz[0]} |d|={d[nz[0]]:.3e} max={d.max():.3e} ndiff={len(nz)}", flush=True)

iters, n, k = 20000, 10000, 32
p = synth_problem(n, k + 1, k, seed=0)
ctx = _lib.Context(0)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
seeds = chain_seeds(11, list(range(4)))
ref, st = ctx.gibbs_run(4, iters, seeds=seeds)
ref2, _ = ctx.gibbs_run(4, iters, seeds=seeds)
first_diff(ref2, ref, "repeat")
print("ref stats", st, flush=True)
for force in (0, 1):
    ctx.set_tuning(cu_limit=128, force_agent_scope=force)
    a, sta = ctx.gibbs_run(2, iters, seeds=seeds[:2])
    print("cu128 force", force, {kk: sta[kk] for kk in ("groups_per_chain", "waves_per_group", "xcd_local_chains", "launches")})
    first_diff(a, ref[:2], f"cu128,force={force}")
ctx.set_tuning(force_agent_scope=1)
a, sta = ctx.gibbs_run(4, iters, seeds=seeds)
first_diff(a, ref, "default geometry, agent scope")
ctx.set_tuning()
import time
with tempfile.TemporaryDirectory() as d:
    t0 = time.time()
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        cu = 128 if rep % 3 != 2 else 0
        pooled, info = T2.run_ranks(pathlib.Path(d), 2, 4, iters, n, k, cu_limit=cu, runs=3)
        bad = not np.array_equal(pooled, ref)
        print(rep, cu, "MISMATCH" if bad else "ok", [(i["loop_ms"], i["xcd_local"], i["unstable_runs"]) for i in info],
              f"{time.time() - t0:.0f}s", flush=True)
        if bad:
            first_diff(pooled, ref, f"  rep {rep}")
            again, _ = ctx.gibbs_run(4, iters, seeds=seeds)
            first_diff(again, ref, "  single process again")
