"""Diagnostic: phase shares of one pass serving several chains (stamped build), wave 0 of group 0."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, "/root/repo")
from pybmc_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libpybmc_amd.so", "libpybmc_amd_stamps.so")
lib = _lib.load_library()
lib.bmc_dev_get_stamps.restype = C.c_int
lib.bmc_dev_get_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
ctx = _lib.Context(0)
rng = np.random.Generator(np.random.PCG64(1))
names = ["u", "B1", "matvec", "wsums", "B2+wavesum", "L1 pub+poll", "s2+record", "top", "L1 sum + L2", "polls"]
T = 2000
for name, n, k, dt, C_ in (("100000x32 f64", 100000, 32, np.float64, 8), ("C4 200000x64 f32", 200000, 64, np.float32, 4),
                           ("C5 50000x256 f64", 50000, 256, np.float64, 8)):
    X = (rng.standard_normal((n, k)) / np.sqrt(n)).astype(dt)
    y = (X.astype(float) @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)).astype(dt)
    ctx.set_problem(y, np.asfortranarray(X), dtype=dt); ctx.set_prior(np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02)
    ctx.gibbs_run(C_, 200, seeds=np.arange(C_) + 1)
    out, st = ctx.gibbs_run(C_, T, seeds=np.arange(C_) + 1)
    buf = (C.c_longlong * 12)()
    lib.bmc_dev_get_stamps(ctx._h, buf)
    cyc = np.array(list(buf), float)[:10] / T
    print(f"{name} chains={C_}: G={st['groups_per_chain']} W={st['waves_per_group']} res={st['residency']} cpp={st['chains_per_pass']} us/iter={st['loop_ms']*1e3/T:.2f} ticks/iter={cyc[:9].sum():.0f}")
    print("   " + "  ".join(f"{n_}:{c:.0f}" for n_, c in zip(names, cyc)))
