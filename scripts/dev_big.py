"""Dev: loop-kernel timing at the C4 / C5 sizes (synthetic gaussian X, not orthogonalised)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from pybmc_amd import _lib
ctx = _lib.Context(0)
rng = np.random.Generator(np.random.PCG64(1))
for name, n, k, dt in (("C4 200000x64 f32", 200000, 64, np.float32), ("C5 50000x256 f64", 50000, 256, np.float64)):
    X = rng.standard_normal((n, k)).astype(dt) / np.sqrt(n)
    beta = rng.standard_normal(k)
    y = (X.astype(float) @ beta + 0.1 * rng.standard_normal(n)).astype(dt)
    t0 = time.time(); ctx.set_problem(y, np.asfortranarray(X), dtype=dt); t1 = time.time()
    ctx.set_prior(np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02); t2 = time.time()
    print(f"{name}: set_problem {t1-t0:.3f}s set_prior {t2-t1:.3f}s")
    b = (n * k + n) * np.dtype(dt).itemsize
    for nch, T in ((1, 2000), (8, 1000)):
        for tune in ((0,0,0,0,0),):
            ctx.set_tuning(*tune)
            ctx.gibbs_run(nch, 100, seeds=np.arange(nch) + 1)
            out, st = ctx.gibbs_run(nch, T, seeds=np.arange(nch) + 1)
            us = st["loop_ms"] * 1e3 / T
            print(f"   chains={nch} G={st['groups_per_chain']} W={st['waves_per_group']} res={st['residency']} launches={st['launches']} "
                  f"us/iter(all chains)={us:.2f} samples/s={nch*T/st['loop_ms']*1e3:.0f} "
                  f"X-pass GB/s={b*nch/us/1e3:.0f} post_ms={st['post_ms']:.2f} rng_ms={st['rng_ms']:.2f} sigma={out[:, T//2:, -1].mean():.4f}")
