"""Round-2 surface additions, on the GPU (-m gpu): the shared per-device context cannot hand one
object another object's problem; train() reaches float32 storage and several devices; planning
for fewer CUs than the device has (and refusing what cannot be resident); the native RCCL
all-gather of the C ABI; the two measurement-only entry points."""
import numpy as np
import pandas as pd
import pytest

from gpu_common import gpu_ctx
from oracle import bmc_oracle as O
from pybmc_amd import BayesianModelCombination, gibbs_sampler
from pybmc_amd import _lib
from pybmc_amd.synthetic import synth_problem

pytestmark = pytest.mark.gpu


def frame(seed, n=400, km=5):
    rng = np.random.Generator(np.random.PCG64(seed))
    truth = rng.standard_normal(n) * 2 + 5
    cols = {"N": np.arange(n), "truth": truth}
    for j in range(km):
        cols[f"m{j}"] = truth + rng.normal(0.2 * j, 1.0, n)
    return pd.DataFrame(cols), [f"m{j}" for j in range(km)]


def make(seed):
    df, models = frame(seed)
    return BayesianModelCombination(models, {"P": df}, "truth"), df


def test_interleaved_objects_sample_their_own_problem():
    """A.orthogonalize(device); B.orthogonalize(device); A.train(): the per-device context is
    shared, so A must notice that the resident problem is no longer its own (same k: nothing
    else would catch it) and upload again.  Reference semantics: every train() samples the
    object's own (y, U_hat) (bmc.py:188-193)."""
    opts = {"iterations": 300, "seeds": [7]}
    alone_a, dfa = make(1)
    alone_a.orthogonalize("P", dfa, 3, method="device")
    alone_a.train(dict(opts))
    alone_b, dfb = make(2)
    alone_b.orthogonalize("P", dfb, 3, method="device")
    alone_b.train(dict(opts))
    assert np.abs(alone_a.samples - alone_b.samples).max() > 1e-3

    a, dfa = make(1)
    b, dfb = make(2)
    a.orthogonalize("P", dfa, 3, method="device")
    b.orthogonalize("P", dfb, 3, method="device")      # replaces A's problem on the device
    a.train(dict(opts))
    assert np.array_equal(a.samples, alone_a.samples)
    b.train(dict(opts))                                  # B's is not resident any more either
    assert np.array_equal(b.samples, alone_b.samples)
    # a functional call in between is just as intrusive
    a.orthogonalize("P", dfa, 3, method="device")
    p = synth_problem(300, 4, 3, seed=5)
    gibbs_sampler(p["y"], p["X"], 50, p["prior"], seeds=[1])
    a.train(dict(opts))
    assert np.array_equal(a.samples, alone_a.samples)
    # and with nothing in between the resident problem IS reused (no second upload)
    a.orthogonalize("P", dfa, 3, method="device")
    gen = _lib.default_context(0).problem_generation
    a.train(dict(opts))
    assert _lib.default_context(0).problem_generation == gen
    assert np.array_equal(a.samples, alone_a.samples)


def test_train_dtype_and_devices_keys():
    a, df = make(3)
    a.orthogonalize("P", df, 3)
    a.train({"iterations": 400, "seeds": [5, 6], "n_chains": 2})
    ref = a.samples.copy()
    assert ref.shape == (800, 4)
    a.train({"iterations": 400, "seeds": [5, 6], "n_chains": 2, "devices": [0]})
    assert np.array_equal(a.samples, ref)
    a.train({"iterations": 400, "seeds": [5, 6], "n_chains": 2, "devices": [0, 0]})   # deduplicated
    assert np.array_equal(a.samples, ref)
    # float32 storage: the same chain up to the rounding of U_hat and y (new capability: 1e-5)
    a.train({"iterations": 400, "seeds": [5, 6], "n_chains": 2, "dtype": "float32"})
    s32 = a.samples
    assert not np.array_equal(s32, ref)
    assert np.abs(s32.mean(0) - ref.mean(0)).max() < 1e-4 * max(1.0, np.abs(ref.mean(0)).max())
    with pytest.raises(ValueError):
        a.train({"iterations": 10, "dtype": "float16"})
    # float32 must not reuse a float64 problem left on the device by orthogonalize
    w32 = s32[:, :-1].mean(0) @ a.Vt_hat          # model weights: free of the SVD's signs
    a.orthogonalize("P", df, 3, method="device")
    gen = _lib.default_context(0).problem_generation
    a.train({"iterations": 400, "seeds": [5, 6], "n_chains": 2, "dtype": np.float32})
    assert _lib.default_context(0).problem_generation == gen + 1      # uploaded again, as float32
    # (the device route signs the singular vectors differently, so the same seeds give a
    # mirrored chain: agreement is at Monte-Carlo level, 800 draws)
    assert np.abs(a.samples[:, :-1].mean(0) @ a.Vt_hat - w32).max() < 1e-2
    assert abs(a.samples[:, -1].mean() - s32[:, -1].mean()) < 1e-2


def test_planning_for_fewer_cus():
    """bmc_tuning.cu_limit: the automatic geometry follows the CUs that may hold a persistent
    launch (a chain that would take 32 CUs of one XCD fits 16 with more panels per wave or
    from LDS) and still reproduces the oracle chain; an explicit request for more groups than
    can be resident is refused at once instead of spinning until the bounded spins expire."""
    ctx = gpu_ctx()
    p = synth_problem(10000, 33, 32, seed=0)
    y, X, prior = p["y"], p["X"], p["prior"]
    ctx.set_problem(y, X)
    ctx.set_prior(*prior)
    T = 100
    st = O.chain_setup(y, X, prior)
    Z, G = O.reference_streams(3, 4, T, 32, O.gamma_shape(st))
    ref, trace = O.gibbs_replay(y, X, T, prior, Z, G, return_sigma2=True)
    W, lam, _ = ctx.basis()
    xi = O.innovations_in_basis(st, y, X, ref, W, lam, trace)
    full, st_full = ctx.gibbs_run(1, T, xi=xi[None], g=G[None])
    assert st_full["groups_per_chain"] == 32
    for limit in (16, 5, 1):
        ctx.set_tuning(cu_limit=limit)
        out, stats = ctx.gibbs_run(1, T, xi=xi[None], g=G[None])
        assert stats["groups_per_chain"] <= limit, (limit, stats)
        assert np.abs(out[0] - ref).max() < 1e-9
    # eight chains on a quarter of the chip: several launches, every chain still its own
    ctx.set_tuning(cu_limit=64)
    many, stats = ctx.gibbs_run(8, T, seeds=np.arange(8) + 1)
    ctx.set_tuning()
    base, _ = ctx.gibbs_run(8, T, seeds=np.arange(8) + 1)
    assert np.abs(many - base).max() < 1e-11
    import time
    ctx.set_tuning(groups_per_chain=32, waves_per_group=5, cu_limit=16)
    t0 = time.perf_counter()
    with pytest.raises(ValueError, match="resident"):
        ctx.gibbs_run(1, T, xi=xi[None], g=G[None])
    assert time.perf_counter() - t0 < 1.0          # refused on the host, no 4 s device spin
    with pytest.raises(ValueError):
        ctx.set_tuning(cu_limit=-1)
    ctx.set_tuning()


def test_native_allgather_world_of_one():
    """bmc_comm_* / bmc_allgather: RCCL straight from the C ABI (no torch.distributed).  One
    GPU here, so the world has one rank; in-place and out-of-place forms."""
    import torch
    ctx = gpu_ctx()
    uid = ctx.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init(1, 0, uid)
    send = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    recv = torch.zeros(1000, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.allgather(send.data_ptr(), recv.data_ptr(), 1000)
    assert torch.equal(send, recv)
    ctx.allgather(recv.data_ptr(), recv.data_ptr(), 1000)          # in place
    assert torch.equal(send, recv)
    # the sampler's output block pooled where it was written
    p = synth_problem(629, 4, 3, seed=3)
    ctx.set_problem(p["y"], p["X"])
    ctx.set_prior(*p["prior"])
    block = torch.empty((2, 100, 4), dtype=torch.float64, device="cuda")
    pooled = torch.empty_like(block)
    ctx.gibbs_run_device(2, 100, [1, 2], block.data_ptr())
    ctx.allgather(block.data_ptr(), pooled.data_ptr(), block.numel())
    host, _ = ctx.gibbs_run(2, 100, seeds=[1, 2])
    assert np.array_equal(pooled.cpu().numpy(), host)
    ctx.comm_destroy()
    with pytest.raises(_lib.BmcError):
        ctx.allgather(send.data_ptr(), recv.data_ptr(), 10)        # no communicator any more
    with pytest.raises(ValueError):
        ctx.comm_init(2, 2, uid)


def test_measurement_entry_points():
    ctx = gpu_ctx()
    p = synth_problem(5000, 17, 16, seed=1)
    ctx.set_problem(p["y"], p["X"])
    want = ctx.gram()
    ms = ctx.gram_bench(reps=5)
    assert 0 < ms < 50 and np.array_equal(ctx.gram(), want)
    rng = np.random.default_rng(0)
    preds = rng.standard_normal((300, 6))
    theta = np.column_stack([rng.standard_normal((4096, 3)) * 0.1, np.ones(4096)])
    ctx.predict(preds, theta, rng.standard_normal((3, 6)), seed=1)
    tm = ctx.predict_timing()
    assert set(tm) == {"h2d_ms", "gemm_ms", "select_ms", "device_ms"}
    assert all(v >= 0 for v in tm.values()) and tm["device_ms"] >= tm["gemm_ms"] > 0


def test_chain_over_xcds_is_the_same_on_every_stream():
    """A chain spread over more than 32 workgroups is organised in teams g mod 8, one per XCD; the
    hardware starts its round robin of workgroups on an XCD that depends on the queue, and the
    kernel renumbers its groups by the XCC id they really run on (kernels_gibbs.hip,
    detect_rotation).  Whatever stream launches it, the chain must be the same bits -- also with a
    group count that is not a multiple of 8 (no renumbering) -- and equal to the oracle replay."""
    import torch
    ctx = gpu_ctx()
    rng = np.random.Generator(np.random.PCG64(21))
    n, k, T = 100000, 32, 150
    X = rng.standard_normal((n, k)) / np.sqrt(n)
    y = X @ rng.standard_normal(k) + 0.1 * rng.standard_normal(n)
    prior = (np.zeros(k), np.eye(k) * 100.0, 1.0, 0.02)
    ctx.set_problem(y, np.asfortranarray(X))
    ctx.set_prior(*prior)
    streams = [torch.cuda.Stream() for _ in range(6)]
    try:
        for G in (0, 80, 79, 256):   # automatic (registers, whole teams), then streamed panels
            ctx.set_tuning(groups_per_chain=G, waves_per_group=8 if G else 0, residency=3 if G else 0)
            first = None
            for s in streams:
                ctx.set_stream(s.cuda_stream)
                out, st = ctx.gibbs_run(2, T, seeds=[5, 6])
                assert st["groups_per_chain"] > 32
                if first is None:
                    first = out
                assert np.array_equal(out, first), (G, st)
            assert np.isfinite(first).all()
            # every group count samples the same posterior (order of summation differs)
            if G == 0:
                ref = first
            else:
                assert np.abs(first - ref).max() < 1e-9
    finally:
        ctx.set_stream(0)
        ctx.set_tuning()
