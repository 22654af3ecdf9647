"""Dev: geometry sweep of the loop kernel at C2 (product build)."""
import sys, numpy as np
sys.path.insert(0, ".")
from pybmc_amd import _lib
from pybmc_amd.synthetic import synth_problem
ctx = _lib.Context(0)
p = synth_problem(10000, 33, 32, 0)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
T = 20000
ref = None
for nch in (1, 8):
    for G, W, res, ppw, agent in [(0,0,0,0,0), (0,0,0,0,1), (20,8,1,1,0), (23,7,1,1,0), (27,6,1,1,0), (32,5,1,1,0), (10,8,1,2,0), (16,5,1,2,0), (32,5,2,0,0), (20,8,2,0,0), (0,0,2,0,0)]:
        try:
            ctx.set_tuning(G, W, res, ppw, agent)
            ctx.gibbs_run(nch, 1000, seeds=np.arange(nch)+1)
            out, st = ctx.gibbs_run(nch, T, seeds=np.arange(nch)+1)
            if ref is None: ref = out[0].copy()
            dev = np.abs(out[0]-ref).max()
            print(f"chains={nch} G={st['groups_per_chain']} W={st['waves_per_group']} res={st['residency']} ppw={ppw} agent={agent} local={st['xcd_local_chains']} us/iter={st['loop_ms']*1e3/T:.3f} samples/s={nch*T/st['loop_ms']*1e3:.0f} maxdev={dev:.2e}")
        except Exception as e:
            print("cfg", G, W, res, ppw, agent, "failed:", e)
