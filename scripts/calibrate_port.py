"""Build-container only (needs /root/reference): time the unmodified reference gibbs_sampler and
the numpy port that bench.py's cpu_baseline uses on the same C2 problem, and print the ratio that
lets the GPU box's cpu_baseline be read as a reference-equivalent."""
import sys, time, numpy as np
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/reference")
from pybmc.inference_utils import gibbs_sampler as ref_gibbs
from oracle import bmc_oracle as O
from pybmc_amd.synthetic import synth_problem

p = synth_problem(10000, 33, 32, 0)
y, X, prior = p["y"], p["X"], p["prior"]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for name, fn in (("reference", lambda: ref_gibbs(y, X, T, list(prior))), ("port", lambda: O.gibbs_port(y, X, T, prior))):
    fn.__call__() if False else None
res = {}
for rep in range(2):
    for name, fn in (("reference", lambda: ref_gibbs(y, X, T, list(prior))),
                     ("port", lambda: O.gibbs_port(y, X, T, prior))):
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
        res.setdefault(name, []).append(T / dt)
for k, v in res.items():
    print(f"{k}: {max(v):.0f} samples/s (best of {len(v)}; C2, {T} iterations)")
print(f"port / reference = {max(res['port']) / max(res['reference']):.3f}")
