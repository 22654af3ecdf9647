#!/usr/bin/env python3
"""Benchmark of the Gibbs hot path: Gibbs samples/s (all chains) at N_obs=10000, K=32.

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete pass of the hot path over the synthetic workload: every
chain of this rank runs ``--iters`` (default 50000) Gibbs iterations -- on-device
variate generation, the persistent loop kernel, the un-rotation of the draws --
and, for N > 1, the one RCCL all-gather that pools the chains.  X, y and the prior
are resident in HBM before the timed region starts.  Weak scaling: one chain per
GPU by default (BASELINE.json configs[1] at N=1, configs[2] at N=8).

Launching.  ``--gpus N`` with N > 1 needs N ranks.  Under ``torch.distributed.run``
(WORLD_SIZE in the environment) this process IS one rank.  Started plainly, it becomes a
parent that never touches the GPU, starts N fresh rank children of itself on 127.0.0.1 and
relays rank 0's JSON line, so ``python bench.py --gpus 8`` works as it stands.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
NOMINAL_CLOCK_GHZ = 2.4      # max shader clock, same guide
F64_VECTOR_TFLOPS = 78.6     # whole chip (256 CUs); one XCD = 1/8
PINGPONG_ONE_WAY_CYCLES = 640   # scripts/micro/pingpong.hip: store -> polled load, one XCD's L2


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--iters", type=int, default=50000)
    ap.add_argument("--chains-per-gpu", type=int, default=1)
    ap.add_argument("--n-obs", type=int, default=10000)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--groups", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--rank-timeout-s", type=float, default=1500.0,
                    help="self-spawned ranks are stopped after this long")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank, so that the "
                         "collective leg runs on a one-GPU box")
    ap.add_argument("--fail-rank", type=int, default=-1,
                    help="test hook: this rank exits with status 3 before the rendezvous (what a "
                         "missing device or a failed RCCL init looks like to the parent)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="rehearse the launch / rendezvous / pooling plumbing on the CPU with "
                         "gloo and a stand-in for the sampler; measures nothing")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------- launching
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv):
    """Parent of a plain ``python bench.py --gpus N``: start N rank children (fresh
    interpreters, one per GPU), relay rank 0's stdout, exit with the worst return code.
    The parent imports neither torch nor the HIP library.

    Fail fast: every child is polled.  As soon as ANY rank exits non-zero (device missing,
    RCCL init error, a sampler status) the others -- which would otherwise sit in the
    rendezvous or in a collective until torch's own timeout -- are killed at once, the failed
    rank and the tail of its output are printed on stderr, and the parent exits non-zero.
    Ranks other than 0 write to per-rank log files (gpurun_out/bench_rank<r>.log), not to
    /dev/null, so their failure text survives."""
    import tempfile
    port = free_port()
    logdir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(logdir, exist_ok=True)
        if not os.access(logdir, os.W_OK):
            raise OSError
    except OSError:
        logdir = tempfile.mkdtemp(prefix="bench_ranks_")
    procs, logs = [], []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # stdout AND stderr of every rank into its own file: nothing blocks on a full pipe
        # while the parent polls, and a dead rank's last words can be shown
        path = os.path.join(logdir, f"bench_rank{r}.log")
        f = open(path, "w+b")
        logs.append((path, f))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv,
                                      env=env, stdout=f, stderr=subprocess.STDOUT, cwd=ROOT))

    def tail(r, n=2500):
        f = logs[r][1]
        f.flush()
        f.seek(0, os.SEEK_END)
        size = f.tell()
        f.seek(max(0, size - n))
        return f.read().decode("utf-8", "replace")

    deadline = time.monotonic() + args.rank_timeout_s
    rc, failed = 0, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = bad[0]
                rc = codes[failed] if codes[failed] > 0 else 1      # (killed by a signal: < 0)
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:      # exactly the children started above, by PID
            if p.poll() is None:
                p.kill()
                p.wait()
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed} exited with status {procs[failed].returncode}; "
                         f"the other ranks were stopped.  Last output of rank {failed} "
                         f"({logs[failed][0]}):\n{tail(failed)}\n")
    elif rc == 124:
        sys.stderr.write(f"bench.py: ranks still running after --rank-timeout-s "
                         f"{args.rank_timeout_s:.0f}; stopped.  Rank 0 said:\n{tail(0)}\n")
    else:
        # rank 0's JSON line (and nothing else of its chatter) is the parent's stdout
        f = logs[0][1]
        f.flush()
        f.seek(0)
        for ln in f.read().decode("utf-8", "replace").splitlines():
            if ln.startswith("{"):
                sys.stdout.write(ln + "\n")
            elif ln.strip():
                sys.stderr.write(ln + "\n")
        sys.stdout.flush()
    for _, f in logs:
        f.close()
    return rc


# ------------------------------------------------------------------------- helpers
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
        names = list(summ["kernels"])
        if summ.get("headline") in summ["kernels"]:     # the headline step's dominant kernel first
            names.remove(summ["headline"])
            names.insert(0, summ["headline"])
        for name in names:
            e = summ["kernels"][name]
            if kernel_substr in name and "hbm_read_bytes_per_launch" in e:
                return {"bytes_per_launch": e["hbm_read_bytes_per_launch"]
                        + e.get("hbm_write_bytes_per_launch", 0.0),
                        "source": os.path.basename(files[-1]), "kernel": name,
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
    # any GPU / CPU ratio is to be read against the FASTER of the two host figures
    out["best_value"] = max(out["value"], out.get("value_1_thread", 0.0))
    return out


class DryRunCtx:
    """--dry-run-cpu only: stands where pybmc_amd._lib.Context stands so that the launch,
    rendezvous and pooling code of this file and of pybmc_amd.chains can be rehearsed on a
    machine without a GPU (gloo).  It samples nothing: every chain's block is filled with a
    function of its seed.  Never used by a measurement."""

    def __init__(self, k):
        import torch
        self.k, self.device, self.torch_device = k, 0, torch.device("cpu")
        self.calls = 0

    def gibbs_run_device(self, n_chains, iters, seeds, out_ptr):
        import ctypes
        n = n_chains * iters * (self.k + 1)
        buf = np.ctypeslib.as_array((ctypes.c_double * n).from_address(out_ptr))
        buf = buf.reshape(n_chains, iters, self.k + 1)
        t = np.arange(iters)[:, None]
        j = np.arange(self.k + 1)[None, :]
        for c, s in enumerate(np.asarray(seeds, dtype=np.uint64)):
            buf[c] = (int(s) % 1000003) * 1e-3 + t + 1e-3 * j + self.calls
        self.calls += 1
        return {"loop_ms": 0.0, "passes": n_chains * iters, "bytes_per_pass": 0,
                "chains_per_pass": 1, "groups_per_chain": 0, "waves_per_group": 0,
                "residency": 0, "xcd_local_chains": 0}


def device_label(torch, idx):
    p = torch.cuda.get_device_properties(idx)
    parts = [p.name]
    if hasattr(p, "pci_bus_id"):
        parts.append("pci %04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id,
                                             getattr(p, "pci_device_id", 0)))
    if hasattr(p, "uuid"):
        parts.append(f"uuid {p.uuid}")
    if hasattr(p, "gcnArchName"):
        parts.append(p.gcnArchName)
    return ", ".join(str(x) for x in parts)


# ------------------------------------------------------------------------- one rank
def rank_main(args):
    import torch  # first: the HIP runtime torch ships must be the one in the process
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    dry = args.dry_run_cpu
    backend = "gloo" if dry else "nccl"
    if args.fail_rank == rank:
        sys.stderr.write(f"rank {rank}: --fail-rank test hook, leaving before the rendezvous\n")
        raise SystemExit(3)
    if not dry and torch.cuda.device_count() < max(world, local_rank + 1):
        # checked BEFORE init_process_group: a rank without a device must not leave the others
        # waiting in the rendezvous (device_count() does not initialise the GPU)
        raise SystemExit(f"rank {rank}: {torch.cuda.device_count()} GPU(s) visible, "
                         f"world size {world} needs one per rank")
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if dry:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    from pybmc_amd.chains import chain_block, chain_seeds, pool_samples

    N, K, T, cpg = args.n_obs, args.k, args.iters, args.chains_per_gpu
    prob = None
    if dry:
        ctx = DryRunCtx(K)
    else:
        from pybmc_amd import _lib
        from pybmc_amd.synthetic import synth_problem
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

    loop_ms, bytes_moved, gather_ms = [], [], []

    def step(record):
        st = ctx.gibbs_run_device(len(mine), T, seeds, out.data_ptr())
        pooled = out
        if use_dist:
            t_g = time.perf_counter()
            pooled = pool_samples(out, n_chains)   # returns once the collective has read `out`
            sync()
            if record:
                gather_ms.append((time.perf_counter() - t_g) * 1e3)
        if record:
            loop_ms.append(st["loop_ms"])
            bytes_moved.append(st["passes"] * st["bytes_per_pass"] / max(st["chains_per_pass"], 1))
        return st, pooled

    for _ in range(args.warmup):
        st, pooled = step(False)
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st, pooled = step(True)
    sync()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # what every rank reports about itself (gathered on rank 0 for the JSON line)
    mine_info = {"rank": rank, "device": "cpu (dry run)" if dry else device_label(torch, local_rank),
                 "loop_ms": float(np.mean(loop_ms)) if loop_ms else 0.0,
                 "allgather_ms": float(np.mean(gather_ms)) if gather_ms else 0.0}
    infos = [mine_info]
    if use_dist:
        infos = [None] * world
        dist.all_gather_object(infos, mine_info)

    if dry:
        # the pooled block must hold every global chain's stand-in values, in chain order
        want = chain_seeds(1, list(range(n_chains)))
        got = pooled[:, 0, 0].numpy() - (ctx.calls - 1)
        assert np.allclose(got, [(int(s) % 1000003) * 1e-3 for s in want]), "pooling order broken"
    else:
        # sanity on the last pooled block: finite and centred on the generating coefficients
        host = pooled[:, T // 5:, :].mean(dim=(0, 1)).cpu().numpy()
        assert np.isfinite(host).all()
        assert abs(host[-1] - 0.1) < 0.01, f"posterior sigma {host[-1]} is off the generating 0.1"

    if rank == 0:
        total_samples = n_chains * T * args.steps
        value = total_samples / elapsed
        avg_loop_ms = float(np.mean(loop_ms))
        line = {
            "metric": "Gibbs samples/sec (all chains) at N_obs=10k, K=32",
            "value": None if dry else value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "none (dry run of the launch plumbing)" if dry else "synthetic",
            "config": {"workload": f"C2: synthetic N_obs={N}, K={K}, f64, {cpg} chain/GPU x {T} "
                                   f"iterations per step, 1 RCCL all-gather per step when N>1",
                       "n_obs": N, "k": K, "iters_per_step": T, "chains_per_gpu": cpg,
                       "parallelism": f"chain-per-gpu x{world}",
                       "groups_per_chain": st["groups_per_chain"],
                       "waves_per_group": st["waves_per_group"],
                       "residency": {1: "vgpr", 2: "lds", 3: "stream"}.get(st["residency"]),
                       "xcd_local_exchange": bool(st["xcd_local_chains"])},
        }
        if dry:
            line["dry_run"] = True
        line["rccl"] = {
            "world": dist.get_world_size() if use_dist else 1,
            "backend": (dist.get_backend() if use_dist else "none (single rank: no collective)"),
            "devices": [i["device"] for i in infos],
            "allgather_bytes_per_rank": len(mine) * T * (K + 1) * 8,
            "allgather_ms": max(i["allgather_ms"] for i in infos),
            "allgather_ms_per_rank": [i["allgather_ms"] for i in infos],
            "loop_ms_min": min(i["loop_ms"] for i in infos),
            "loop_ms_max": max(i["loop_ms"] for i in infos),
        }
        if not dry:
            line["roofline"] = roofline_entry(st, avg_loop_ms, float(np.mean(bytes_moved)), N, K, T)
            if world == 1 and not args.no_extra:
                line["extra"] = extras(ctx, torch, dev, local_rank, N, K, T)
                # the kernel BASELINE.json's north_star names for the HBM roofline (>= 40 %): the
                # residual reduction at the C4 size, as a roofline block of its own
                rc4 = line["extra"].pop("residual_rss_c4", None)
                if rc4 and "error" not in rc4:
                    line["roofline_residual_c4"] = residual_roofline(rc4)
            if world == 1 and not args.no_cpu_baseline:
                cb = cpu_baseline(prob, budget_s=args.cpu_budget_s)
                line["cpu_baseline"] = cb
                line["gpu_over_cpu_best"] = value / cb["best_value"]
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def roofline_entry(st, avg_loop_ms, alg_bytes, N, K, T):
    """The dominant kernel's line.  ``achieved`` is SURVEY 8(d)'s algorithmic bytes
    ((N*K+N)*8 per iteration x iterations) over the loop kernel's HIP-event time.  What binds
    the kernel depends on where the panels live: streamed from HBM -> "hbm"; resident in
    VGPRs / LDS -> the per-iteration dependency chain ("latency"): the bytes are then never
    moved, and ``frac`` is NOT an HBM utilisation -- the latency figures below are what to read."""
    achieved = alg_bytes / (avg_loop_ms * 1e-3) / 1e9
    residency = {1: "vgpr", 2: "lds", 3: "stream"}.get(st["residency"])
    us_it = avg_loop_ms * 1e3 / T
    tr = profiled_traffic("gibbs_loop_kernel") or {}
    traffic = tr.get("bytes_per_launch")
    # the counters come from a committed profile of an EARLIER run of this command: say how far
    # that run's kernel time is from the one just measured, so that stale traffic shows
    prof_ms = tr.get("profiled_avg_ms")
    stale = (abs(prof_ms - avg_loop_ms) / avg_loop_ms > 0.05) if prof_ms else None
    r = {"bound": "hbm" if residency == "stream" else "latency", "kernel": "gibbs_loop_kernel",
         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_source": tr.get("source"),
         "profiled_kernel": tr.get("kernel"), "profiled_avg_ms": prof_ms, "traffic_stale": stale,
         "algorithmic_bytes_per_launch": alg_bytes,
         "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
         "loop_ms_per_launch": avg_loop_ms, "us_per_iteration": us_it, "residency": residency}
    if residency != "stream":
        xcds = 1 if st["xcd_local_chains"] else 8
        flops_it = 2.0 * N * K + 3.0 * N
        r["latency"] = {
            "cycles_per_iteration_at_2.4GHz": us_it * NOMINAL_CLOCK_GHZ * 1e3,
            "exchange_floor_cycles_one_way": PINGPONG_ONE_WAY_CYCLES,
            "exchange_floor_source": "scripts/micro/pingpong.hip (store -> polled load through one XCD's L2)",
            "xcds_used": xcds,
            "f64_fma_frac_of_xcds_used": flops_it / (us_it * 1e-6) / 1e12 / (F64_VECTOR_TFLOPS * xcds / 8),
        }
        r["note"] = ("row panels stay in " + residency.upper() + " for the whole launch: HBM traffic "
                     "is the variates in and the draws out; the bound is the chain of dependent "
                     "steps of one iteration (draw -> residual pass -> wave/group/cross-CU sums -> "
                     "sigma2), see DESIGN.md 4.1")
    return r


def residual_roofline(rc4):
    """residual_rss_kernel at C4 (N = 200 000, K = 64, f32: 52 MB per pass) in the shape of the
    `roofline` block: live HIP-event time per launch, counter traffic from the committed profile,
    and how far that profile's kernel time is from the live one."""
    tr = profiled_traffic("residual_rss_kernel<float, 2, 1>") or {}
    ms = rc4["ms_per_pass"]
    prof_ms = tr.get("profiled_avg_ms")
    traffic = tr.get("bytes_per_launch")
    return {"bound": "hbm", "kernel": "residual_rss_kernel<float, 2, 1>",
            "workload": "C4: N_obs=200000, K=64, f32 storage, one coefficient vector per pass",
            "achieved": rc4["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": rc4["frac_of_8TBs"], "ms_per_launch": ms,
            "algorithmic_bytes_per_launch": rc4["bytes_per_pass"],
            "traffic": traffic, "traffic_source": tr.get("source"),
            "traffic_over_algorithmic": (traffic / rc4["bytes_per_pass"]) if traffic else None,
            "profiled_kernel": tr.get("kernel"), "profiled_avg_ms": prof_ms,
            # (a 10 us kernel moves by +- 10 % from box to box and run to run: the counters are
            # called stale only when the profiled time is more than 25 % off)
            "traffic_stale": (abs(prof_ms - ms) / ms > 0.25) if prof_ms else None,
            "ms_per_launch_runs": rc4.get("ms_per_pass_runs"),
            "served_from": rc4["served_from"], "note": rc4["note"]}


def e2e_surface(local_rank, iters=50000):
    """Wall clock of the REFERENCE SURFACE: BayesianModelCombination.orthogonalize -> train
    (50 000 iterations) -> predict2 -> evaluate, the way docs/usage.md drives the reference
    (pybmc/bmc.py:79-376), with everything a user pays for: host SVD / pandas, uploads, the host
    K x K algebra of set_prior, the loop, the copy back of the samples and of rndm_m.  Beside
    each, the pieces timed on their own through the same context, and the reference's figures
    measured in the build container (other CPU, read as orders of
    magnitude)."""
    import contextlib
    import io

    import pandas as pd

    from pybmc_amd import BayesianModelCombination, _lib

    def tick():
        return time.perf_counter()

    out = {}
    for tag, n, km, k in (("c1_629x4_k3", 629, 4, 3), ("c2_shaped_10000x33_k32", 10000, 33, 32)):
        rng = np.random.Generator(np.random.PCG64(0))
        truth = rng.standard_normal(n) * 2 + 5
        cols = {"N": np.arange(n), "Z": np.arange(n) % 97, "truth": truth}
        models = [f"m{j}" for j in range(km)]
        for j, name in enumerate(models):
            cols[name] = truth + rng.normal(0.1 * j, 1.0, n) + rng.normal(0, 0.5, n)
        df = pd.DataFrame(cols)
        bmc = BayesianModelCombination(models, {"P": df}, "truth", device=local_rank)
        e = {}
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):       # ([INFO] default-value prints, as the reference)
            for rep in range(2):                     # second pass timed: buffers exist, clocks up
                t0 = tick(); bmc.orthogonalize("P", df, k); e["orthogonalize_s"] = tick() - t0
                t0 = tick(); bmc.train({"iterations": iters}); e["train_s"] = tick() - t0
                t0 = tick(); rndm_m, lo, med, up = bmc.predict2("P"); e["predict2_s"] = tick() - t0
                t0 = tick(); cov = bmc.evaluate(); e["evaluate_s"] = tick() - t0
        e["total_s"] = e["orthogonalize_s"] + e["train_s"] + e["predict2_s"] + e["evaluate_s"]
        e["train_samples_per_s"] = iters / e["train_s"]
        st = bmc.last_stats
        e["train_device_ms"] = {"variates": st["rng_ms"], "loop": st["loop_ms"], "unrotate": st["post_ms"]}
        e["rndm_m"] = {"shape": list(rndm_m.shape), "c_contiguous": bool(rndm_m.flags.c_contiguous),
                       "MB": rndm_m.nbytes / 1e6}
        assert len(cov) == 21 and med.shape[0] == n
        # the pieces of train() on their own
        ctx = _lib.default_context(local_rank)
        y, X = bmc.centered_experiment_train, bmc.U_hat
        t0 = tick(); ctx.set_problem(y, X); up_s = tick() - t0
        t0 = tick(); ctx.set_prior(np.zeros(k), np.diag(bmc.S_hat ** 2), 1.0, 0.02); pr_s = tick() - t0
        runs = []
        for rep in range(3):      # (one call at a time is noisy on the host side: best of three)
            t0 = tick(); smp, st2 = ctx.gibbs_run(1, iters, seeds=[3]); runs.append(tick() - t0)
        run_s = min(runs)
        e["train_split_s"] = {"upload_panelize_gram": up_s, "set_prior_host_algebra_rotate": pr_s,
                              "variates_loop_unrotate_copyback": run_s,
                              "variates_loop_unrotate_copyback_runs": runs,
                              "of_which_loop": st2["loop_ms"] * 1e-3,
                              "copyback_MB": smp.nbytes / 1e6,
                              "python_and_prints": max(0.0, e["train_s"] - up_s - pr_s - run_s)}
        out[tag] = e
    out["c1_629x4_k3"]["reference_cpu"] = ("train() 8 117 samples/s at N=629, K_models=15, kept 3 "
                                            "(BASELINE.md section 2, 8 vCPU Xeon)")
    out["c2_shaped_10000x33_k32"]["reference_cpu"] = (
        "gibbs_sampler 1.1-1.6 k samples/s at N=10 000, k=31; orthogonalize (full-matrices SVD) 6.6 s; "
        "predict2 3.7 s at M=2 000 (BASELINE.md section 2)")
    # the C5 posterior predictive through the context: bands + coverage only, and with the
    # 4 GB of draws returned in the reference's layout (C-ordered (10000, M): device transpose +
    # one contiguous copy into pageable numpy memory)
    rng = np.random.Generator(np.random.PCG64(55))
    Mp, Kmp, kp, Sp = 50000, 257, 256, 10000
    preds = rng.standard_normal((Mp, Kmp))
    Vt_hat = rng.standard_normal((kp, Kmp)) * 0.05
    theta = np.column_stack([rng.standard_normal((Sp, kp)) * 0.1, rng.uniform(0.05, 0.15, Sp)])
    cp = _lib.Context(local_rank)
    e = {}
    for want, key in ((False, "bands_and_coverage_only_s"), (True, "with_rndm_m_returned_s")):
        for rep in range(2):
            t0 = tick()
            r = cp.predict(preds, theta, Vt_hat, seed=9, truth=preds.mean(1),
                           cov_percentiles=list(range(0, 101, 5)), want_draws=want)
            e[key] = tick() - t0
        if want:
            e["rndm_m"] = {"shape": list(r[0].shape), "c_contiguous": bool(r[0].flags.c_contiguous),
                           "GB": r[0].nbytes / 1e9}
        del r
    tm = cp.predict_timing()
    e["device_ms"] = tm
    e["reference_cpu"] = "rndm_m_random_calculator 11.7 s at M=5 000, K=32 (BASELINE.md section 2)"
    out["c5_predictive_50000x257"] = e
    cp.close()
    return out


def extras(ctx, torch, dev, local_rank, N, K, T):
    from pybmc_amd import _lib
    from pybmc_amd.chains import chain_seeds
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
        # 16 / 32 / 64 chains: the register-resident panels of an XCD serve a bundle of 2 / 4 / 8
        # chains per pass, one bundle per XCD, ONE launch (every chain bit-identical to its solo
        # run: tests/test_gibbs_parity_gpu.py::test_bundles_of_chains_per_xcd)
        for nc in (16, 32, 64):
            seedsn = chain_seeds(1, list(range(nc)))
            outn = torch.empty((nc, T, K + 1), dtype=torch.float64, device=dev)
            ctx.gibbs_run_device(nc, T, seedsn, outn.data_ptr())
            t1 = time.perf_counter()
            stn = ctx.gibbs_run_device(nc, T, seedsn, outn.data_ptr())
            torch.cuda.synchronize()
            en = time.perf_counter() - t1
            extra[f"chains{nc}_one_gpu"] = {"samples_per_s": nc * T / en, "loop_ms": stn["loop_ms"],
                                            "us_per_iteration_all": stn["loop_ms"] * 1e3 / T,
                                            "launches": stn["launches"],
                                            "chains_per_pass": stn["chains_per_pass"],
                                            "waves_per_group": stn["waves_per_group"]}
            del outn
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
        # the reference's own sizes (BASELINE configs[0]: 629 rows; a few thousand rows): one wave,
        # or 2 / 4 waves of one workgroup, per chain (gibbs_wave_kernel) -- the loop alone
        small = {}
        for tag, n_s, k_s in (("c1_n629_k3", 629, 3), ("n2500_k8", 2500, 8)):
            rng_s = np.random.Generator(np.random.PCG64(n_s))
            Xs = rng_s.standard_normal((n_s, k_s)) / np.sqrt(n_s)
            ys = Xs @ rng_s.standard_normal(k_s) + 0.1 * rng_s.standard_normal(n_s)
            ctx.set_problem(ys, np.asfortranarray(Xs))
            ctx.set_prior(np.zeros(k_s), np.eye(k_s) * 10.0, 1.0, 0.02)
            row = {}
            for cs in (1, 256):
                Ts = 50000 if cs == 1 else 5000
                outs = torch.empty((cs, Ts, k_s + 1), dtype=torch.float64, device=dev)
                ss = chain_seeds(1, list(range(cs)))
                ctx.gibbs_run_device(cs, Ts, ss, outs.data_ptr())
                sts = ctx.gibbs_run_device(cs, Ts, ss, outs.data_ptr())
                row[f"chains{cs}"] = {"us_per_iteration_all": sts["loop_ms"] * 1e3 / Ts,
                                      "samples_per_s": cs * Ts / (sts["loop_ms"] * 1e-3),
                                      "waves_per_chain": sts["waves_per_group"],
                                      "groups_per_chain": sts["groups_per_chain"]}
                del outs
            small[tag] = row
        small["reference_cpu"] = "train() 8 117 samples/s at N=629 (BASELINE.md section 2)"
        extra["reference_sized_problems"] = small
    except Exception as e:
        extra["reference_sized_problems"] = {"error": str(e)}
    try:
        # residual-reduction kernel at the C4 size (N=200000, K=64, f32 storage)
        rng = np.random.Generator(np.random.PCG64(4))
        X4 = np.asfortranarray(rng.standard_normal((200000, 64), dtype=np.float32))
        y4 = rng.standard_normal(200000, dtype=np.float32)
        c4 = _lib.Context(local_rank)
        c4.set_problem(y4, X4, dtype=np.float32)
        # (best of three runs of 50 back-to-back launches: a 10 us kernel right behind a 52 MB
        # upload sees the clock still ramping; the spread is reported)
        runs4 = [c4.residual_rss_bench(nb=1, reps=50) for _ in range(3)]
        ms = min(runs4)
        b4 = (200000 * 64 + 200000) * 4
        extra["residual_rss_c4"] = {"ms_per_pass": ms, "achieved_GBs": b4 / (ms * 1e-3) / 1e9,
                                    "frac_of_8TBs": b4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "bytes_per_pass": b4, "served_from": "infinity cache",
                                    "ms_per_pass_runs": runs4,
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
                                     "bytes_per_pass": bb, "served_from": "hbm",
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
            residency = {1: "vgpr", 2: "lds", 3: "stream"}.get(stl["residency"])
            e = {"n_obs": n4, "k": k4, "dtype": "f32" if dt == np.float32 else "f64",
                 "chains8_us_per_iteration_all": st8l["loop_ms"] * 1e3 / it8,
                 "chains8_samples_per_s": 8 * it8 / st8l["loop_ms"] * 1e3,
                 "chains8_per_pass": st8l["chains_per_pass"],
                 "us_per_iteration": us, "samples_per_s": iters4 / stl["loop_ms"] * 1e3,
                 "algorithmic_GBs": bl / us / 1e3, "residency": residency,
                 "bound": "hbm" if residency == "stream" and bl > 256 * 2 ** 20 else
                          "infinity cache" if residency == "stream" else "latency",
                 "groups": stl["groups_per_chain"], "waves": stl["waves_per_group"]}
            if residency == "stream":   # bytes really move every iteration: an HBM-roofline fraction
                e["frac_of_8TBs"] = bl / us / 1e3 / HBM_PEAK_GBS
            else:                        # resident panels: bytes never move, no bandwidth fraction
                e["algorithmic_bytes_over_time_vs_8TBs"] = bl / us / 1e3 / HBM_PEAK_GBS
            extra[tag] = e
            if tag == "loop_c5":
                # the one-off f64 MFMA Gram at the C5 shape (set-up row a1)
                ms = cl.gram_bench(reps=20)
                ka_pad = -(-(k4 + 1) // 16) * 16
                npairs = (ka_pad // 16) * (ka_pad // 16 + 1) // 2       # upper-triangle tiles
                fl_exec = npairs * 2.0 * 256 * (-(-n4 // 64) * 64)      # what the MFMAs execute
                extra["gram_c5"] = {"ms_per_launch_pair": ms, "flops_executed": fl_exec,
                                    "flops_full_square": 2.0 * n4 * (k4 + 1) ** 2,
                                    "TFLOPs_executed": fl_exec / (ms * 1e-3) / 1e12,
                                    "frac_of_78.6TF_f64_matrix_peak": fl_exec / (ms * 1e-3) / 1e12 / 78.6,
                                    "bytes_read_algorithmic": bl, "GBs": bl / (ms * 1e-3) / 1e9,
                                    "note": "[X y]'[X y] with v_mfma_f64_16x16x4_f64, upper-triangle "
                                            "16x16 tiles only (mirrored by the reduce kernel); time = "
                                            "gram_mfma_kernel + gram_reduce_kernel"}
            del outl, Xl, yl
            cl.close()
    except Exception as e:
        extra["loop_c4"] = {"error": str(e)}
    try:
        # BASELINE configs[4]'s posterior-predictive leg at full size: 10000 draws x 50000
        # held-out points x 257 models (reference sampling_utils.py:57-82), device generator,
        # bands + coverage, draws left on the device (what evaluate() / predict bands need)
        rng = np.random.Generator(np.random.PCG64(55))
        Mp, Kmp, kp, Sp = 50000, 257, 256, 10000
        preds = rng.standard_normal((Mp, Kmp))
        Vt_hat = rng.standard_normal((kp, Kmp)) * 0.05
        theta = np.column_stack([rng.standard_normal((Sp, kp)) * 0.1, rng.uniform(0.05, 0.15, Sp)])
        truth = preds.mean(1)
        cp = _lib.Context(local_rank)
        for _ in range(2):
            cp.predict(preds, theta, Vt_hat, seed=9, truth=truth,
                       cov_percentiles=list(range(0, 101, 5)), want_draws=False)
        tm = cp.predict_timing()
        fl = 2.0 * Sp * Kmp * Mp
        extra["predict_c5"] = {"points": Mp, "models": Kmp, "draws": Sp,
                               "gemm_ms": tm["gemm_ms"], "order_stat_ms": tm["select_ms"],
                               "device_ms": tm["device_ms"], "h2d_ms": tm["h2d_ms"],
                               "gemm_TFLOPs": fl / (tm["gemm_ms"] * 1e-3) / 1e12,
                               "gemm_frac_of_78.6TF_f64_matrix_peak":
                                   fl / (tm["gemm_ms"] * 1e-3) / 1e12 / 78.6,
                               "draws_written_GB": Mp * Sp * 8 / 1e9,
                               "reference_cpu_s_at_M5000_K32": 11.7}
        cp.close()
    except Exception as e:
        extra["predict_c5"] = {"error": str(e)}
    try:
        extra["e2e"] = e2e_surface(local_rank)
    except Exception as e:
        extra["e2e"] = {"error": repr(e)}
    return extra


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args, argv))
    rank_main(args)


if __name__ == "__main__":
    main()
