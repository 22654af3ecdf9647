#!/usr/bin/env python3
"""Benchmark of the Gibbs hot path: Gibbs samples/s (all chains) at N_obs=10000, K=32.

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete pass of the hot path over the synthetic workload: every
chain of this rank runs ``--iters`` (default 50000) Gibbs iterations -- on-device
variate generation, the persistent loop kernel, the un-rotation of the draws --
and, for N > 1, the one RCCL all-gather that pools the chains.  X, y and the prior
are resident in HBM before the timed region starts.  Weak scaling: one chain per
GPU by default (BASELINE.json configs[1] at N=1, configs[2] at N=8).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def profiled_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 PMC summary
    (profiles/rNN_summary.json, FETCH_SIZE/WRITE_SIZE collected in separate passes and
    corrected as the MI355X guide prescribes).  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            summ = json.load(f)
        for name, e in summ["kernels"].items():
            if kernel_substr in name and "hbm_read_bytes_per_launch" in e:
                return {"bytes_per_launch": e["hbm_read_bytes_per_launch"]
                        + e.get("hbm_write_bytes_per_launch", 0.0),
                        "source": os.path.basename(files[-1]),
                        "profiled_avg_ms": e["avg_ms"]}
    except Exception:
        return None
    return None


def cpu_baseline(problem, budget_s=12.0, chunk=1000):
    """Time the numpy port of the reference loop (oracle/bmc_oracle.gibbs_port: same
    numpy calls per iteration as reference inference_utils.py:39-54) on the host for
    about ``budget_s`` seconds of the same C2 workload."""
    from oracle import bmc_oracle as O
    y, X, prior = problem["y"], problem["X"], problem["prior"]
    O.gibbs_port(y, X, 20, prior)  # warm BLAS
    done, t0 = 0, time.perf_counter()
    while True:
        O.gibbs_port(y, X, chunk, prior)
        done += chunk
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    threads = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        info = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        if info:
            threads = int(info[0]["num_threads"])
    except Exception:
        pass
    out = {"value": done / el, "unit": "samples/s", "cores": threads, "kind": "port",
           "sample": f"{done} iterations of 1 chain (N={X.shape[0]}, K={X.shape[1]}, f64), numpy port of "
                     f"reference gibbs_sampler, {el:.1f} s, BLAS threads={threads} of "
                     f"{os.cpu_count()} host cpus"}
    # the same loop on one BLAS thread (SURVEY.md 8d asks for both figures), a shorter sample
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1, user_api="blas"):
            O.gibbs_port(y, X, 20, prior)
            d1, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < budget_s / 3:
                O.gibbs_port(y, X, chunk, prior)
                d1 += chunk
            out["value_1_thread"] = d1 / (time.perf_counter() - t1)
    except Exception:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--iters", type=int, default=50000)
    ap.add_argument("--chains-per-gpu", type=int, default=1)
    ap.add_argument("--n-obs", type=int, default=10000)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--groups", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    args = ap.parse_args()

    import torch  # first: the HIP runtime torch ships must be the one in the process
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pybmc_amd import _lib
    from pybmc_amd.chains import chain_block, chain_seeds, pool_samples
    from pybmc_amd.synthetic import synth_problem

    N, K, T, cpg = args.n_obs, args.k, args.iters, args.chains_per_gpu
    prob = synth_problem(N, K + 1, K, seed=0)
    ctx = _lib.Context(local_rank)
    ctx.set_problem(prob["y"], prob["X"])              # X, y -> HBM (outside the timed region)
    ctx.set_prior(*prob["prior"])
    if args.groups or args.waves:
        ctx.set_tuning(args.groups, args.waves)
    n_chains = world * cpg
    mine = chain_block(n_chains, world, rank)
    seeds = chain_seeds(1, mine)
    out = torch.empty((len(mine), T, K + 1), dtype=torch.float64, device=dev)

    loop_ms, bytes_moved = [], []

    def step(record):
        st = ctx.gibbs_run_device(len(mine), T, seeds, out.data_ptr())
        pooled = out
        if world > 1:
            pooled = pool_samples(out, n_chains)
            torch.cuda.synchronize()   # the all-gather has read `out` before the next step rewrites it
        if record:
            loop_ms.append(st["loop_ms"])
            bytes_moved.append(st["passes"] * st["bytes_per_pass"] / max(st["chains_per_pass"], 1))
        return st, pooled

    for _ in range(args.warmup):
        st, pooled = step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st, pooled = step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # sanity on the last pooled block: finite and centred on the generating coefficients
    host = pooled[:, T // 5:, :].mean(dim=(0, 1)).cpu().numpy()
    assert np.isfinite(host).all()
    assert abs(host[-1] - 0.1) < 0.01, f"posterior sigma {host[-1]} is off the generating 0.1"

    if rank == 0:
        total_samples = n_chains * T * args.steps
        value = total_samples / elapsed
        avg_loop_ms = float(np.mean(loop_ms))
        achieved = float(np.mean(bytes_moved)) / (avg_loop_ms * 1e-3) / 1e9
        line = {
            "metric": "Gibbs samples/sec (all chains) at N_obs=10k, K=32",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C2: synthetic N_obs={N}, K={K}, f64, {cpg} chain/GPU x {T} "
                                   f"iterations per step, 1 RCCL all-gather per step when N>1",
                       "n_obs": N, "k": K, "iters_per_step": T, "chains_per_gpu": cpg,
                       "parallelism": f"chain-per-gpu x{world}",
                       "groups_per_chain": st["groups_per_chain"],
                       "waves_per_group": st["waves_per_group"],
                       "residency": {1: "vgpr", 2: "lds", 3: "stream"}.get(st["residency"]),
                       "xcd_local_exchange": bool(st["xcd_local_chains"])},
            "roofline": {"bound": "hbm", "kernel": "gibbs_loop_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (profiled_traffic("gibbs_loop_kernel") or {}).get(
                             "bytes_per_launch"),
                         "traffic_source": (profiled_traffic("gibbs_loop_kernel") or {}).get("source"),
                         "algorithmic_bytes_per_launch": float(np.mean(bytes_moved)),
                         "note": "achieved = algorithmic bytes ((N*K+N)*8 per iteration x "
                                 "iterations) / HIP-event time of the loop kernel; at this size "
                                 "the row panels stay in VGPRs for the whole launch, so the HBM "
                                 "traffic (variates in, draws out) is far below the algorithmic "
                                 "bytes and the binding limit is per-iteration latency (DESIGN.md)",
                         "loop_ms_per_launch": avg_loop_ms,
                         "us_per_iteration": avg_loop_ms * 1e3 / T},
        }
        if world == 1 and not args.no_extra:
            extra = {}
            try:
                seeds8 = chain_seeds(1, list(range(8)))
                out8 = torch.empty((8, T, K + 1), dtype=torch.float64, device=dev)
                ctx.gibbs_run_device(8, T, seeds8, out8.data_ptr())
                t1 = time.perf_counter()
                st8 = ctx.gibbs_run_device(8, T, seeds8, out8.data_ptr())
                torch.cuda.synchronize()
                e8 = time.perf_counter() - t1
                extra["chains8_one_gpu"] = {"samples_per_s": 8 * T / e8, "loop_ms": st8["loop_ms"],
                                            "groups_per_chain": st8["groups_per_chain"]}
                del out8
                # 16 chains: two per XCD, still one launch
                seeds16 = chain_seeds(1, list(range(16)))
                out16 = torch.empty((16, T, K + 1), dtype=torch.float64, device=dev)
                ctx.gibbs_run_device(16, T, seeds16, out16.data_ptr())
                t1 = time.perf_counter()
                st16 = ctx.gibbs_run_device(16, T, seeds16, out16.data_ptr())
                torch.cuda.synchronize()
                e16 = time.perf_counter() - t1
                extra["chains16_one_gpu"] = {"samples_per_s": 16 * T / e16, "loop_ms": st16["loop_ms"],
                                             "launches": st16["launches"]}
                del out16
            except Exception as e:  # never lose the headline line
                extra["chains8_one_gpu"] = {"error": str(e)}
            try:
                # opt-in, reported separately: rss from sufficient statistics
                # instead of the per-iteration pass over the data that the headline measures
                ctx.set_tuning(rss_mode=1)
                og = {}
                for cg in (1, 256):
                    outg = torch.empty((cg, T, K + 1), dtype=torch.float64, device=dev)
                    sg = chain_seeds(1, list(range(cg)))
                    ctx.gibbs_run_device(cg, T, sg, outg.data_ptr())
                    stg = ctx.gibbs_run_device(cg, T, sg, outg.data_ptr())
                    og[f"chains{cg}"] = {"samples_per_s": cg * T / (stg["loop_ms"] * 1e-3),
                                         "us_per_iteration": stg["loop_ms"] * 1e3 / T}
                    del outg
                og["note"] = ("NOT the headline path: no pass over X inside the loop "
                              "(rss(u) = rss(u0) - 2 d'g0 + d'Gd, K <= 64), one wave per chain")
                extra["opt_in_rss_from_sufficient_statistics"] = og
            except Exception as e:
                extra["opt_in_rss_from_sufficient_statistics"] = {"error": str(e)}
            finally:
                ctx.set_tuning()
            try:
                # residual-reduction kernel at the C4 size (N=200000, K=64, f32 storage)
                rng = np.random.Generator(np.random.PCG64(4))
                X4 = np.asfortranarray(rng.standard_normal((200000, 64), dtype=np.float32))
                y4 = rng.standard_normal(200000, dtype=np.float32)
                c4 = _lib.Context(local_rank)
                c4.set_problem(y4, X4, dtype=np.float32)
                ms = c4.residual_rss_bench(nb=1, reps=50)
                b4 = (200000 * 64 + 200000) * 4
                extra["residual_rss_c4"] = {"ms_per_pass": ms, "achieved_GBs": b4 / (ms * 1e-3) / 1e9,
                                            "frac_of_8TBs": b4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "bytes_per_pass": b4,
                                            "note": "52 MB working set: served by the 256 MiB "
                                                    "Infinity Cache after the first pass"}
                del X4, y4
                # the same kernel on a working set that cannot stay in the Infinity Cache
                nbig = 2_000_000
                Xb = np.asfortranarray(rng.standard_normal((nbig, 64), dtype=np.float32))
                yb = rng.standard_normal(nbig, dtype=np.float32)
                c4.set_problem(yb, Xb, dtype=np.float32)
                ms = c4.residual_rss_bench(nb=1, reps=20)
                bb = (nbig * 64 + nbig) * 4
                extra["residual_rss_hbm"] = {"n_obs": nbig, "k": 64, "dtype": "f32",
                                             "ms_per_pass": ms,
                                             "achieved_GBs": bb / (ms * 1e-3) / 1e9,
                                             "frac_of_8TBs": bb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "bytes_per_pass": bb,
                                             "note": "520 MB per pass > 256 MiB Infinity Cache: HBM-served"}
                del Xb, yb
                c4.close()
            except Exception as e:
                extra["residual_rss_c4"] = {"error": str(e)}
            try:
                # the persistent loop at the C4 and C5 sizes (1 chain, as on each GPU of the
                # 8-GPU configurations); gaussian design matrices scaled to unit-norm columns
                # ... and on a matrix larger than the 256 MiB Infinity Cache (410 MB): the streaming
                # loop against HBM itself
                for tag, n4, k4, dt, iters4 in (("loop_c4", 200000, 64, np.float32, 4000),
                                                ("loop_c5", 50000, 256, np.float64, 2000),
                                                ("loop_hbm", 400000, 256, np.float32, 500)):
                    rng = np.random.Generator(np.random.PCG64(8))
                    Xl = rng.standard_normal((n4, k4), dtype=np.float32)
                    Xl *= np.float32(1.0 / np.sqrt(n4))
                    Xl = Xl.astype(dt, copy=False)
                    yl = (Xl @ rng.standard_normal(k4).astype(dt)
                          + 0.1 * rng.standard_normal(n4)).astype(dt)
                    cl = _lib.Context(local_rank)
                    cl.set_problem(yl, np.asfortranarray(Xl), dtype=dt)
                    cl.set_prior(np.zeros(k4), np.eye(k4) * 100.0, 1.0, 0.02)
                    outl = torch.empty((1, iters4, k4 + 1), dtype=torch.float64, device=dev)
                    cl.gibbs_run_device(1, 200, chain_seeds(1, [0]), outl.data_ptr())
                    stl = cl.gibbs_run_device(1, iters4, chain_seeds(1, [0]), outl.data_ptr())
                    bl = stl["bytes_per_pass"]
                    us = stl["loop_ms"] * 1e3 / iters4
                    # the configuration's 8 chains on this one GPU (BASELINE C4 / C5 name 8 chains)
                    it8 = max(200, iters4 // 4)
                    out8l = torch.empty((8, it8, k4 + 1), dtype=torch.float64, device=dev)
                    cl.gibbs_run_device(8, 100, chain_seeds(1, list(range(8))), out8l.data_ptr())
                    st8l = cl.gibbs_run_device(8, it8, chain_seeds(1, list(range(8))), out8l.data_ptr())
                    del out8l
                    extra[tag] = {"n_obs": n4, "k": k4, "dtype": "f32" if dt == np.float32 else "f64",
                                  "chains8_us_per_iteration_all": st8l["loop_ms"] * 1e3 / it8,
                                  "chains8_samples_per_s": 8 * it8 / st8l["loop_ms"] * 1e3,
                                  "chains8_per_pass": st8l["chains_per_pass"],
                                  "us_per_iteration": us, "samples_per_s": iters4 / stl["loop_ms"] * 1e3,
                                  "algorithmic_GBs": bl / us / 1e3,
                                  "frac_of_8TBs": bl / us / 1e3 / HBM_PEAK_GBS,
                                  "residency": {1: "vgpr", 2: "lds", 3: "stream"}.get(stl["residency"]),
                                  "groups": stl["groups_per_chain"], "waves": stl["waves_per_group"]}
                    del outl, Xl, yl
                    cl.close()
            except Exception as e:
                extra["loop_c4"] = {"error": str(e)}
            line["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(prob)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
