"""Diagnostic: phase shares of one Gibbs iteration (stamped build, never the product)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
from pybmc_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libpybmc_amd.so", "libpybmc_amd_stamps.so")
from pybmc_amd.synthetic import synth_problem
lib = _lib.load_library()
lib.bmc_dev_get_stamps.restype = C.c_int
lib.bmc_dev_get_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
ctx = _lib.Context(0)
p = synth_problem(10000, 33, 32, 0)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
names = ["u", "B1", "matvec", "wsum", "B2+wavesum", "pub+poll", "s2", "top", "gsum", "polls"]
T = 20000
cfgs = [(0, 0, 0, 0, 0), (32, 5, 1, 1, 0), (32, 4, 1, 2, 0), (20, 8, 1, 1, 0)]
for G, W, res, ppw, agent in cfgs:
    ctx.set_tuning(G, W, res, ppw, agent)
    ctx.gibbs_run(1, 2000, seeds=[1])
    out, st = ctx.gibbs_run(1, T, seeds=[1])
    buf = (C.c_longlong * 12)()
    lib.bmc_dev_get_stamps(ctx._h, buf)
    cyc = np.array(list(buf), float)[:10] / T
    print(f"G={G} W={W} res={st['residency']} ppw={ppw} local={st['xcd_local_chains']} us/iter={st['loop_ms']*1e3/T:.3f} ticks/iter={cyc[:9].sum():.0f}")
    print("   " + "  ".join(f"{n}:{c:.0f}" for n, c in zip(names, cyc)))
