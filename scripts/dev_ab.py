"""Dev: A/B of loop schedules at C2 in ONE process, interleaved rounds."""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpu_common import golden_case, gpu_ctx, replay_inputs
ctx = gpu_ctx()
g, y, X, prior = golden_case("gibbs_c2_10000x32")
ctx.set_problem(y, X); ctx.set_prior(*prior)
st, xi, ref = replay_inputs(ctx, g, y, X, prior, 500)
variants = {"leader G32W5": dict(groups_per_chain=32, waves_per_group=5), "allpoll G32W5": dict(groups_per_chain=32, waves_per_group=5, schedule=99),
            "leader G20W8": dict(groups_per_chain=20, waves_per_group=8), "allpoll G20W8": dict(groups_per_chain=20, waves_per_group=8, schedule=99),
            "allpoll G40W4": dict(groups_per_chain=40, waves_per_group=4, schedule=99)}
for name, kw in variants.items():
    ctx.set_tuning(**kw)
    out, stats = ctx.gibbs_run(1, 500, xi=xi[None], g=g["G"][None, :500])
    print(name, "replay err", np.abs(out[0] - ref).max(), "local", stats["xcd_local_chains"], "G", stats["groups_per_chain"], "W", stats["waves_per_group"])
T = 20000
res = {k: [] for k in variants}
for rnd in range(5):
    for name, kw in variants.items():
        ctx.set_tuning(**kw)
        for nch in (1,):
            _, s = ctx.gibbs_run(nch, T, seeds=np.arange(nch) + 1)
            res[name].append(s["loop_ms"] * 1e3 / T)
for name, v in res.items():
    print(f"{name:16s} us/iter median {np.median(v):.3f} min {min(v):.3f}")
