"""Dev: small problems (notebook-sized): default geometry, 1 / 8 / 256 chains."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
from pybmc_amd.synthetic import synth_problem
ctx = _lib.Context(0)
T = 20000
for n, km, k in ((629, 4, 3), (629, 16, 15), (1000, 33, 32), (2000, 32, 31), (400, 6, 5), (64, 3, 2)):
    p = synth_problem(n, km, k, 3)
    ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
    for nch in (1, 8, 256):
        ctx.gibbs_run(nch, 500, seeds=np.arange(nch) + 1)
        out, st = ctx.gibbs_run(nch, T, seeds=np.arange(nch) + 1)
        print(f"N={n} K={k} chains={nch}: G={st['groups_per_chain']} W={st['waves_per_group']} res={st['residency']} launches={st['launches']} {st['loop_ms']*1e3/T:.3f} us/iter  {nch*T/st['loop_ms']*1e3:.0f} samples/s  sigma={out[:, T//2:, -1].mean():.4f}")
