import sys, time, numpy as np
sys.path.insert(0, ".")
from pybmc_amd import _lib
from oracle import bmc_oracle as O
from tests.conftest import load_golden
from pybmc_amd.synthetic import synth_problem

ctx = _lib.Context(0)
for name in ["gibbs_tiny3x2", "gibbs_dense64x8", "gibbs_ortho629x3", "gibbs_ragged1237x5", "gibbs_c2_10000x32"]:
    g = load_golden(name)
    if "X" in g: X, y = g["X"], g["y"]
    else:
        p = synth_problem(10000, 33, 32, 0); X, y = p["X"], p["y"]
    prior = (g["b0"], g["C0"], float(g["nu0"]), float(g["s20"]))
    ctx.set_problem(y, X); ctx.set_prior(*prior)
    K = X.shape[1]
    Ga = ctx.gram()
    Xa = np.column_stack([np.asarray(X, float), y])
    print(name, "gram err", np.abs(Ga - Xa.T @ Xa).max() / np.abs(Xa.T@Xa).max())
    W, lam, s2i = ctx.basis()
    st = O.chain_setup(y, np.asarray(X,float), prior)
    print("  s2 init", s2i, st["sigma2_init"], abs(s2i-st["sigma2_init"])/s2i)
    m, c = ctx.conditional_moments(0.37)
    mo, co = O.conditional_moments(st, y, np.asarray(X,float), 0.37)
    print("  moments err", np.abs(m-mo).max()/np.abs(mo).max(), np.abs(c-co).max()/np.abs(co).max())
    b = np.random.default_rng(0).standard_normal((3, K))
    r = ctx.residual_rss(b)
    ro = [O.residual_rss(y, np.asarray(X,float), bb) for bb in b]
    print("  rss err", np.abs(r-ro).max()/np.abs(ro).max())
    T = int(g["T"])
    samples = g["samples"]
    _, tr = O.gibbs_replay(y, np.asarray(X,float), T, prior, g["Z"], g["G"], return_sigma2=True) if name != "gibbs_c2_10000x32" else (None, None)
    xi = O.innovations_in_basis(st, y, np.asarray(X,float), samples, W, lam, tr)
    out, stats = ctx.gibbs_run(1, T, xi=xi[None], g=g["G"][None])
    err = np.abs(out[0]-samples).max()
    pm = np.abs(out[0].mean(0)-samples.mean(0)).max()/np.abs(samples.mean(0)).max()
    print("  chain maxabs err", err, "post-mean rel err", pm, stats)
# perf
p = synth_problem(10000, 33, 32, 0)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
for G, Wv in [(0,0),(40,4),(157,1),(79,2),(32,5),(20,8),(16,10)]:
    ctx.set_tuning(G, Wv)
    out, stats = ctx.gibbs_run(1, 20000, seeds=[1])
    print("C2 1 chain", G, Wv, "loop_ms", stats["loop_ms"], "us/iter", stats["loop_ms"]*1e3/20000, "rng", stats["rng_ms"], "post", stats["post_ms"], "res", stats["residency"], stats["groups_per_chain"], stats["waves_per_group"])
ctx.set_tuning(0,0)
out, stats = ctx.gibbs_run(8, 20000, seeds=np.arange(1,9))
print("C2 8 chains", stats)
print("beta mean", out[:, 2000:, :3].mean((0,1)), p["beta_true"][:3], "sigma", out[:, 2000:, -1].mean())
