"""Dev: residual-reduction kernel bandwidth at C4 and HBM scale for several grid caps."""
import os, sys, numpy as np
sys.path.insert(0, ".")
from pybmc_amd import _lib
rng = np.random.Generator(np.random.PCG64(4))
X4 = np.asfortranarray(rng.standard_normal((200000, 64), dtype=np.float32)); y4 = rng.standard_normal(200000, dtype=np.float32)
Xb = np.asfortranarray(rng.standard_normal((2000000, 64), dtype=np.float32)); yb = rng.standard_normal(2000000, dtype=np.float32)
Xd = np.asfortranarray(rng.standard_normal((1000000, 64))); yd = rng.standard_normal(1000000)
for cap in (2048, 1280, 1024, 768, 512, 256):
    os.environ["BMC_RSS_GROUPS_CAP"] = str(cap)
    c = _lib.Context(0)
    for name, X, y, dt in (("C4 52MB f32", X4, y4, np.float32), ("HBM 520MB f32", Xb, yb, np.float32), ("HBM 520MB f64", Xd, yd, np.float64)):
        c.set_problem(y, X, dtype=dt)
        b = X.size * X.itemsize + y.size * y.itemsize
        for nb in (1, 8):
            ms = c.residual_rss_bench(nb=nb, reps=30)
            print(f"cap={cap} {name} nb={nb}: {ms*1e3:.1f} us/pass {b/ms/1e6:.0f} GB/s ({b/ms/1e6/8000:.2%})")
    # correctness spot check
    beta = rng.standard_normal((2, 64))
    got = c.residual_rss(beta); want = [float(np.sum((yd - Xd @ bb) ** 2)) for bb in beta]
    print("   check rel err", np.abs(got - want).max() / max(want))
    c.close()
