"""Dev: residual-reduction kernel bandwidth at 52 MB (Infinity-Cache resident) and 520 MB."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
_lib.LIB_PATH = sys.argv[1] if len(sys.argv) > 1 else _lib.LIB_PATH
rng = np.random.Generator(np.random.PCG64(4))
X4 = np.asfortranarray(rng.standard_normal((200000, 64), dtype=np.float32)); y4 = rng.standard_normal(200000, dtype=np.float32)
Xb = np.asfortranarray(rng.standard_normal((2000000, 64), dtype=np.float32)); yb = rng.standard_normal(2000000, dtype=np.float32)
Xd = np.asfortranarray(rng.standard_normal((1000000, 64))); yd = rng.standard_normal(1000000)
c = _lib.Context(0)
for name, X, y, dt in (("C4 52MB f32", X4, y4, np.float32), ("HBM 520MB f32", Xb, yb, np.float32), ("HBM 520MB f64", Xd, yd, np.float64)):
    c.set_problem(y, X, dtype=dt)
    b = X.size * X.itemsize + y.size * y.itemsize
    for nb in (1, 8):
        ms = c.residual_rss_bench(nb=nb, reps=30)
        print(f"{_lib.LIB_PATH.split('/')[-1]} {name} nb={nb}: {ms*1e3:.1f} us/pass {b/ms/1e6:.0f} GB/s ({b/ms/1e6/8000:.2%})", flush=True)
beta = rng.standard_normal((2, 64))
got = c.residual_rss(beta); want = [float(np.sum((yd - Xd @ bb) ** 2)) for bb in beta]
print("   check rel err", np.abs(got - want).max() / max(want))
