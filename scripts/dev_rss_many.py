import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
ctx = _lib.Context(0)
rng = np.random.default_rng(0)
n, k = 200000, 64
X = rng.standard_normal((n, k)).astype(np.float32); y = rng.standard_normal(n).astype(np.float32)
ctx.set_problem(y, np.asfortranarray(X), dtype=np.float32)
B = rng.standard_normal((203, k))
ctx.residual_rss(B[:8])
t0 = time.time(); got = ctx.residual_rss(B); dt = time.time() - t0
want = ((y.astype(float)[None, :] - B @ X.astype(float).T) ** 2).sum(1)
print("203 vectors at C4 size:", f"{dt*1e3:.2f} ms", "max rel err", np.abs(got - want).max() / want.max())
