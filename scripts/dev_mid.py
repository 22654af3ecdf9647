"""Dev: mid-size problems (K between 65 and 256, LDS residency) and many-chain throughput."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
ctx = _lib.Context(0)
rng = np.random.Generator(np.random.PCG64(1))
T = 3000
for n, k in ((30000, 32), (60000, 16), (20000, 32), (16000, 64), (30000, 48)):
    X = rng.standard_normal((n, k)) / np.sqrt(n)
    y = X @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)
    ctx.set_problem(y, np.asfortranarray(X)); ctx.set_prior(np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02)
    for nch in (1, 8):
        ctx.gibbs_run(nch, 200, seeds=np.arange(nch) + 1)
        out, st = ctx.gibbs_run(nch, T, seeds=np.arange(nch) + 1)
        b = (n * k + n) * 8
        us = st["loop_ms"] * 1e3 / T
        print(f"N={n} K={k} chains={nch}: G={st['groups_per_chain']} W={st['waves_per_group']} res={st['residency']} cpp={st['chains_per_pass']} launches={st['launches']} local={st['xcd_local_chains']} {us:.2f} us/iter(all) {nch*T/st['loop_ms']*1e3:.0f} samples/s  alg {b*nch/us/1e3:.0f} GB/s sigma={out[:, T//2:, -1].mean():.4f}")
