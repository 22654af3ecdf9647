"""Independent chains across GPUs: one process per GPU, one all-gather at the end.

The reference runs ONE chain per ``train()`` call (bmc.py:188-193) and has no
notion of devices; multi-chain / multi-GPU sampling is a capability of this build
(SURVEY.md section 8e).  Chains are independent given (X, y, prior), so rank r
runs its block of chains with zero communication and the per-rank sample blocks
``[chains_r, T, k+1]`` are pooled by a single ``all_gather`` (RCCL over xGMI with
backend "nccl"; gloo in the CPU tests).  Seeds are assigned per GLOBAL chain id,
so the pooled posterior does not depend on the number of ranks.
"""
from __future__ import annotations

import contextlib

import numpy as np

SEED_STRIDE = 0x9E3779B97F4A7C15  # golden-ratio increment between chain keys


def chain_block(n_chains, world_size, rank):
    """Global chain ids owned by ``rank``: contiguous blocks, sizes differ by at most 1."""
    if n_chains < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad chain partition arguments")
    base, extra = divmod(n_chains, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def max_block(n_chains, world_size):
    return -(-n_chains // world_size)


def chain_seeds(base_seed, chain_ids):
    """Philox key of each chain: a function of (base_seed, global chain id) only."""
    ids = np.asarray(chain_ids, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return (np.uint64(base_seed & 0xFFFFFFFFFFFFFFFF)
                + (ids + np.uint64(1)) * np.uint64(SEED_STRIDE))


def pool_samples(local_block, n_chains, group=None):
    """All-gather the per-rank blocks into ``[n_chains, T, k+1]`` (a torch tensor on
    the same device).  ``local_block`` is ``[len(chain_block(...)), T, k+1]``.  Ranks
    may own unequal numbers of chains: blocks are padded to the largest one for the
    collective and the padding is dropped afterwards."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return local_block
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = chain_block(n_chains, world, rank)
    if local_block.shape[0] != len(mine):
        raise ValueError("local block does not match this rank's chain block")
    mb = max_block(n_chains, world)
    T, k1 = local_block.shape[1], local_block.shape[2]
    send = local_block
    if len(mine) != mb:
        send = torch.zeros((mb, T, k1), dtype=local_block.dtype, device=local_block.device)
        send[:len(mine)] = local_block
    pooled = torch.empty((world * mb, T, k1), dtype=local_block.dtype,
                         device=local_block.device)
    dist.all_gather_into_tensor(pooled, send.contiguous(), group=group)
    if local_block.is_cuda:
        # RCCL runs the collective on its own stream and torch only makes ITS current stream
        # wait for it.  The sampler writes `local_block` on the library's own non-blocking
        # stream, which is ordered against neither: wait here, on the host, until the
        # collective has read `send` (= `local_block` itself when the blocks are even), so the
        # caller may hand the same buffer to the next run.
        torch.cuda.current_stream(local_block.device).synchronize()
    if n_chains == world * mb:
        return pooled
    keep = []
    for r in range(world):
        nr = len(chain_block(n_chains, world, r))
        keep.append(pooled[r * mb:r * mb + nr])
    return torch.cat(keep, dim=0)


def posterior_summary(samples, Vt_hat=None, burn=0):
    """Pooled posterior summaries used by the parity tests and the bench: means and
    variances of beta, mean sigma and sigma^2, and (with Vt_hat) the posterior mean
    model weights ``mean(beta) @ Vt_hat + 1/K`` (reference sampling_utils.py:64-67)."""
    s = np.asarray(samples)
    s = s.reshape(-1, s.shape[-1]) if s.ndim == 2 else s[:, burn:].reshape(-1, s.shape[-1])
    beta, sig = s[:, :-1], s[:, -1]
    out = dict(beta_mean=beta.mean(0), beta_var=beta.var(0), sigma_mean=sig.mean(),
               sigma2_mean=(sig ** 2).mean())
    if Vt_hat is not None:
        out["weights_mean"] = out["beta_mean"] @ Vt_hat + 1.0 / Vt_hat.shape[1]
    return out


def run_chains(ctx, n_chains, iterations, base_seed=0, group=None, out=None):
    """Run this rank's share of ``n_chains`` chains on its GPU (problem and prior
    already set on ``ctx``) and pool them.  Returns (pooled torch tensor, stats)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    mine = chain_block(n_chains, world, rank)
    # (a context may name the torch device its buffers live on; a bmc_ctx is always a GPU)
    dev = getattr(ctx, "torch_device", None) or torch.device("cuda", ctx.device)
    if out is None:
        out = torch.empty((len(mine), iterations, ctx.k + 1), dtype=torch.float64, device=dev)
    elif tuple(out.shape) != (len(mine), iterations, ctx.k + 1) or out.dtype != torch.float64 \
            or not out.is_contiguous():
        raise ValueError("out must be a contiguous float64 [chains of this rank, iterations, k+1]")
    stats = None
    if mine:
        stats = ctx.gibbs_run_device(len(mine), iterations, chain_seeds(base_seed, mine),
                                     out.data_ptr())
    return pool_samples(out, n_chains, group), stats


def run_on_devices(y, X, iterations, prior_info, n_chains, seeds, devices, dtype=None):
    """Several GPUs driven from ONE process (what ``train(devices=[...])`` uses): the chains
    are split over the devices like ranks split them (``chain_block``), each device gets its own
    context and host thread (the C ABI releases the GIL and a bmc_ctx is single-threaded), and
    the blocks are concatenated in global chain order.  A device listed twice is used once: two
    persistent launches on one GPU would compete for the same CUs.
    Returns (samples [iterations, k+1] or [n_chains, iterations, k+1], list of per-device stats)."""
    import threading

    from . import _lib
    from .inference_utils import _draw_seeds

    devs = list(dict.fromkeys(int(d) for d in devices))
    if not devs:
        raise ValueError("devices must name at least one GPU")
    if seeds is None:
        seeds = _draw_seeds(n_chains)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64).reshape(n_chains)
    blocks = [chain_block(n_chains, len(devs), i) for i in range(len(devs))]
    results, errors = [None] * len(devs), [None] * len(devs)
    b0, C0, nu0, s20 = prior_info
    ctxs = [_lib.default_context(d) for d in devs]

    def work(i):
        try:
            if not blocks[i]:
                return
            ctx = ctxs[i]
            # (the per-device context is shared with the functional API; a stand-in context of
            # the CPU tests has no lock)
            with getattr(ctx, "lock", None) or contextlib.nullcontext():
                ctx.set_problem(y, X, dtype=dtype)
                ctx.set_prior(b0, C0, nu0, s20)
                results[i] = ctx.gibbs_run(len(blocks[i]), int(iterations), seeds=seeds[blocks[i]])
        except BaseException as e:     # re-raised on the calling thread
            errors[i] = e

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(devs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    out = np.concatenate([r[0] for r in results if r is not None], axis=0)
    stats = [r[1] for r in results if r is not None]
    return (out[0] if n_chains == 1 else out), stats
