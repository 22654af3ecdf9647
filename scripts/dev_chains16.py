"""Dev: many chains at the C2 size -- 8 (one per XCD) against 16 and 32 (two per XCD)."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
from pybmc_amd.synthetic import synth_problem
ctx = _lib.Context(0)
p = synth_problem(10000, 33, 32, 0)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
T = 20000
ref, _ = ctx.gibbs_run(1, T, seeds=[5])
for C in (1, 8, 16, 24, 32):
    seeds = np.arange(C) + 1
    ctx.gibbs_run(C, 2000, seeds=seeds)
    out, st = ctx.gibbs_run(C, T, seeds=seeds)
    same = np.array_equal(out[4], ref[0]) if C > 4 else None
    print(C, "chains:", {k: st[k] for k in ("launches", "groups_per_chain", "waves_per_group", "xcd_local_chains")},
          f"{st['loop_ms']*1e3/T:.3f} us/iter  {C*T/st['loop_ms']/1e3:.2f} M samples/s  chain(seed 5) identical to solo: {same}", flush=True)

