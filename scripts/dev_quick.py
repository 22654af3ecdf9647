"""Dev: C2 quick check -- replay parity on the 10000x32 golden + timing of 1 and 8 chains."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from gpu_common import golden_case, gpu_ctx, replay_inputs
ctx = gpu_ctx()
g, y, X, prior = golden_case("gibbs_c2_10000x32")
ctx.set_problem(y, X); ctx.set_prior(*prior)
st, xi, ref = replay_inputs(ctx, g, y, X, prior, 500)
out, stats = ctx.gibbs_run(1, 500, xi=xi[None], g=g["G"][None, :500])
print("replay max abs err", np.abs(out[0] - ref).max())
T = 20000
for nch in (1, 8):
    ctx.gibbs_run(nch, 1000, seeds=np.arange(nch) + 1)
    best = min(ctx.gibbs_run(nch, T, seeds=np.arange(nch) + 1)[1]["loop_ms"] for _ in range(3))
    print(f"chains={nch}: {best*1e3/T:.3f} us/iter, {nch*T/best*1e3:.0f} samples/s")
