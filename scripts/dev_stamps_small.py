"""Diagnostic: phase shares for a single-workgroup chain (no exchange), stamped build."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libpybmc_amd.so", "libpybmc_amd_stamps.so")
from pybmc_amd.synthetic import synth_problem
lib = _lib.load_library()
lib.bmc_dev_get_stamps.restype = C.c_int
lib.bmc_dev_get_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
ctx = _lib.Context(0)
names = ["u", "B1", "matvec", "-", "-", "wsum..rss", "s2", "top"]
T = 20000
for n, km, k, tune in ((629, 4, 3, dict(groups_per_chain=1, waves_per_group=4)), (629, 4, 3, dict(groups_per_chain=1, waves_per_group=1, panels_per_wave=0)),
                       (64, 4, 3, dict(groups_per_chain=1, waves_per_group=1))):
    p = synth_problem(n, km, k, 3)
    ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
    try:
        ctx.set_tuning(**tune)
        ctx.gibbs_run(1, 500, seeds=[1])
        out, st = ctx.gibbs_run(1, T, seeds=[1])
    except Exception as e:
        print(n, k, tune, e); continue
    buf = (C.c_longlong * 12)()
    lib.bmc_dev_get_stamps(ctx._h, buf)
    cyc = np.array(list(buf), float) / T
    print(f"N={n} K={k} G={st['groups_per_chain']} W={st['waves_per_group']} res={st['residency']} us/iter={st['loop_ms']*1e3/T:.3f} ticks/iter={cyc.sum():.0f}")
    print("   " + "  ".join(f"{n_}:{c:.0f}" for n_, c in zip(names, cyc)))
