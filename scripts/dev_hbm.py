import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
ctx = _lib.Context(0)
rng = np.random.Generator(np.random.PCG64(1))
for name, n, k, dt in (("C5 50000x256 f64", 50000, 256, np.float64), ("400000x256 f32 (410 MB)", 400000, 256, np.float32), ("300000x200 f64 (482 MB)", 300000, 200, np.float64)):
    X = (rng.standard_normal((n, k)) / np.sqrt(n)).astype(dt)
    y = (X.astype(float) @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)).astype(dt)
    ctx.set_problem(y, np.asfortranarray(X), dtype=dt); ctx.set_prior(np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02)
    ctx.gibbs_run(1, 200, seeds=[1])
    for C in (1, 8):
        v = []
        for _ in range(3):
            out, st = ctx.gibbs_run(C, 600, seeds=np.arange(C) + 1); v.append(st["loop_ms"] / 600 * 1e3)
        b = st["bytes_per_pass"]
        print(name, C, "chains: G", st["groups_per_chain"], "res", st["residency"], "cpp", st["chains_per_pass"], f"{min(v):.2f} us/iter = {b/min(v)/1e6:.2f} TB/s", "sigma", round(float(out[..., 200:, -1].mean()), 4), flush=True)
