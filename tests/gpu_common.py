"""Shared helpers of the -m gpu tests."""
import numpy as np

from conftest import load_golden
from oracle import bmc_oracle as O
from pybmc_amd.synthetic import synth_problem, sha256

_ctx = None


def gpu_ctx():
    """One context for the whole test session (one process on the GPU)."""
    global _ctx
    if _ctx is None:
        from pybmc_amd import _lib
        _ctx = _lib.Context(0)
    _ctx.set_tuning(0, 0, 0, 0)
    return _ctx


def golden_case(name):
    g = load_golden(name)
    if "X" in g:
        X, y = g["X"], g["y"]
    else:
        n, km, kept, seed = (int(v) for v in g["synth"])
        p = synth_problem(n, km, kept, seed)
        X, y = p["X"], p["y"]
        assert sha256(np.asarray(X, float, order="F")) == str(g["X_sha"])
    prior = (g["b0"], g["C0"], float(g["nu0"]), float(g["s20"]))
    return g, y, X, prior


def replay_inputs(ctx, g, y, X, prior, T):
    """Innovations of the reference's chain expressed in the library's basis."""
    Xf = np.asarray(X, float)
    st = O.chain_setup(y, Xf, prior)
    W, lam, s2i = ctx.basis()
    ref = g["samples"][:T]
    # exact sigma2 trace: sigma2_t from the golden G stream and the golden betas
    trace = np.empty(T + 1)
    trace[0] = st["sigma2_init"]
    for t in range(T):
        trace[t + 1] = O.sigma2_draw(st, O.residual_rss(y, Xf, ref[t, :-1]), g["G"][t])
    xi = O.innovations_in_basis(st, y, Xf, ref, W, lam, trace)
    return st, xi, ref
