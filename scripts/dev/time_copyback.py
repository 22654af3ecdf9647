"""Where does train()'s time beyond the loop go?  gibbs_run (samples copied to a fresh numpy
array) against gibbs_run_device (samples stay on the GPU), C2, 50 000 iterations."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pybmc_amd import _lib
from pybmc_amd.synthetic import synth_problem

p = synth_problem(10000, 33, 32, seed=0)
ctx = _lib.Context(0)
ctx.set_problem(p["y"], p["X"]); ctx.set_prior(*p["prior"])
T = 50000
out_d = torch.empty((1, T, 33), dtype=torch.float64, device="cuda:0")
for rep in range(4):
    t0 = time.perf_counter(); st = ctx.gibbs_run_device(1, T, [3], out_d.data_ptr()); t1 = time.perf_counter()
    t2 = time.perf_counter(); out, st2 = ctx.gibbs_run(1, T, seeds=[3]); t3 = time.perf_counter()
    t4 = time.perf_counter(); buf = np.empty((1, T, 33)); buf[:] = 0.0; t5 = time.perf_counter()
    print(f"rep {rep}: device-out {1e3*(t1-t0):.2f} ms (loop {st['loop_ms']:.2f}, total_ms {st['total_ms']:.2f}) | "
          f"host-out {1e3*(t3-t2):.2f} ms (loop {st2['loop_ms']:.2f}) | fresh 13.2 MB alloc+touch {1e3*(t5-t4):.2f} ms", flush=True)
