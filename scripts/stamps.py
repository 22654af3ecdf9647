#!/usr/bin/env python3
"""Phase shares of one Gibbs iteration from the stamped diagnostic build
(`make -C pybmc_amd/csrc stamps` -> pybmc_amd/libpybmc_amd_stamps.so, never loaded by the
package): s_memtime stamps taken by wave 0 of group 0 of chain 0.  Read the SHARES, not the
length -- the stamps and their scheduling fences add ~13 %.

    python scripts/stamps.py [case ...]      cases: c2 (default) n100k c4 c5 small
    python scripts/stamps.py c2 --geo 32,5,1,1 --geo 20,8,1,1     (groups,waves,residency,ppw)
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pybmc_amd import _lib  # noqa: E402
from pybmc_amd.synthetic import synth_problem  # noqa: E402

NAMES = ["u", "B1", "pass", "wave sums", "B2 + sum of waves", "publish -> gathered", "sigma2", "loop top",
         "sum of groups (+ level 2)", "polls"]


def problem(case):
    if case in ("c2", "small"):
        n, k = (10000, 32) if case == "c2" else (629, 3)
        p = synth_problem(n, k + 1, k, seed=0)
        return p["y"], p["X"], p["prior"], np.float64
    n, k, dt = {"n100k": (100000, 32, np.float64), "c4": (200000, 64, np.float32),
                "c5": (50000, 256, np.float64)}[case]
    rng = np.random.Generator(np.random.PCG64(1))
    X = (rng.standard_normal((n, k)) / np.sqrt(n)).astype(dt)
    y = (X.astype(float) @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)).astype(dt)
    return y, np.asfortranarray(X), (np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02), dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=["c2"])
    ap.add_argument("--geo", action="append", default=[], help="groups,waves,residency,panels_per_wave")
    ap.add_argument("--chains", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20000)
    args = ap.parse_args()
    _lib._share_hip_runtime_with_torch()
    lib = _lib.bind(_lib.LIB_PATH.replace("libpybmc_amd.so", "libpybmc_amd_stamps.so"))
    lib.bmc_dev_get_stamps.restype = C.c_int
    lib.bmc_dev_get_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    geos = [tuple(int(v) for v in g.split(",")) for g in args.geo] or [(0, 0, 0, 0)]
    for case in args.cases:
        y, X, prior, dt = problem(case)
        ctx = _lib.Context(0, lib=lib)
        ctx.set_problem(y, X, dtype=dt)
        ctx.set_prior(*prior)
        T = args.iters if case in ("c2", "small") else max(500, args.iters // 10)
        for geo in geos:
            ctx.set_tuning(*geo)
            seeds = np.arange(args.chains) + 1
            ctx.gibbs_run(args.chains, max(100, T // 10), seeds=seeds)
            _, st = ctx.gibbs_run(args.chains, T, seeds=seeds)
            buf = (C.c_longlong * 12)()
            lib.bmc_dev_get_stamps(ctx._h, buf)
            cyc = np.array(list(buf), float)[:10] / T
            print(f"{case} geo={geo} chains={args.chains}: G={st['groups_per_chain']} W={st['waves_per_group']} "
                  f"res={st['residency']} cpp={st['chains_per_pass']} local={st['xcd_local_chains']} "
                  f"us/iter={st['loop_ms'] * 1e3 / T:.3f} ticks/iter={cyc[:9].sum():.0f}")
            print("   " + "  ".join(f"{n}:{c:.0f}" for n, c in zip(NAMES, cyc)), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
