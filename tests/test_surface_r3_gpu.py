"""Round-3 surface additions, on the GPU (-m gpu): rndm_m in the reference's memory layout, the
draws fetched after the fact, the width limits as ValueErrors that name the limit, and the RCCL
loader failing with a message instead of a crash."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gpu_common import gpu_ctx
from pybmc_amd import _lib, gibbs_sampler, rndm_m_random_calculator

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def predictive_inputs(M=173, Km=7, k=5, S=10000, seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    preds = rng.standard_normal((M, Km))
    Vt = rng.standard_normal((k, Km)) * 0.1
    theta = np.column_stack([rng.standard_normal((S, k)) * 0.2, rng.uniform(0.05, 0.2, S)])
    return preds, theta, Vt


def test_rndm_m_has_the_reference_layout():
    """The reference returns a C-ordered (10000, M) array (sampling_utils.py:77:
    `noiseless + standard_normal(...) * sigma`); so does this build -- same values as the device
    layout, transposed on the device (ragged M and S: partial 64 x 64 tiles)."""
    ctx = gpu_ctx()
    for M, S in ((173, 10000), (64, 2048), (1, 777)):
        preds, theta, Vt = predictive_inputs(M=M, S=S)
        c_arr, bands, _ = ctx.predict(preds, theta, Vt, seed=4)
        assert c_arr.shape == (S, M) and c_arr.flags.c_contiguous
        f_arr = ctx.predict_draws("F")
        assert f_arr.shape == (S, M) and f_arr.flags.f_contiguous
        assert np.array_equal(c_arr, f_arr)
        assert np.array_equal(bands, np.percentile(c_arr, (2.5, 50, 97.5), axis=0))
        # bands / coverage only, the draws fetched afterwards: the same array
        _, bands2, _ = ctx.predict(preds, theta, Vt, seed=4, want_draws=False)
        assert np.array_equal(bands2, bands) and np.array_equal(ctx.predict_draws(), c_arr)
    g = np.random.Generator(np.random.PCG64(1))
    samples = np.column_stack([g.standard_normal((12000, 5)) * 0.2, g.uniform(0.05, 0.2, 12000)])
    preds, _, Vt = predictive_inputs()
    rndm_m, (lo, med, up) = rndm_m_random_calculator(preds, samples, Vt, seed=8)
    assert rndm_m.shape == (10000, 173) and rndm_m.flags.c_contiguous and rndm_m.flags.owndata


def test_predict_draws_needs_a_predict():
    ctx = _lib.Context(0)
    with pytest.raises(_lib.BmcError):
        ctx.predict_draws()
    ctx.close()


def test_width_limits_raise_value_errors_that_name_the_limit():
    """The reference is plain numpy and accepts any width (inference_utils.py:25,41;
    sampling_utils.py:57 fixes the draws at 10000).  This build's limits -- listed in
    INTEGRATION.md section 6 -- surface as ValueError with the limit in the message, never as a
    wrong result: K <= 256 columns in the sampler, <= 255 models on the device orthogonalize
    route, <= 16384 predictive draws, <= 64 percentiles / coverage intervals."""
    rng = np.random.Generator(np.random.PCG64(0))
    X = rng.standard_normal((400, 257))
    y = rng.standard_normal(400)
    with pytest.raises(ValueError, match="256"):
        gibbs_sampler(y, X, 10, (np.zeros(257), np.eye(257), 1.0, 0.02))
    # 256 columns is inside the limit and runs
    out = gibbs_sampler(y, X[:, :256], 5, (np.zeros(256), np.eye(256), 1.0, 0.02), seeds=[1])
    assert out.shape == (5, 257) and np.isfinite(out).all()
    ctx = gpu_ctx()
    preds, theta, Vt = predictive_inputs(M=8, S=20000)
    with pytest.raises(ValueError, match="16384"):
        ctx.predict(preds, theta, Vt, seed=1)
    preds, theta, Vt = predictive_inputs(M=8, S=16384)
    draws, bands, _ = ctx.predict(preds, theta, Vt, seed=1)
    assert np.array_equal(bands, np.percentile(draws, (2.5, 50, 97.5), axis=0))
    with pytest.raises(ValueError, match="64"):
        ctx.predict(preds, theta, Vt, seed=1, q=tuple(np.linspace(1, 99, 65)))
    with pytest.raises(ValueError, match="255"):
        ctx.orthogonalize(rng.standard_normal((600, 256)), rng.standard_normal(600), 3)


def test_comm_init_without_rccl_reports_the_loader_error():
    """bmc_comm_init when RCCL cannot be loaded: BMC_EHIP with the loader's text (round-2
    advisor finding: the message was built from a NULL dlerror() and crashed)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from pybmc_amd import _lib\n"
            "ctx = _lib.Context(0)\n"
            "for _ in range(2):\n"
            "    try:\n"
            "        ctx.comm_init(1, 0, bytes(128))\n"
            "        print('no error')\n"
            "    except _lib.BmcError as e:\n"
            "        print('BmcError:', e)\n" % ROOT)
    env = dict(os.environ, BMC_RCCL_SONAME="libdoes_not_exist_bmc.so.9")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("BmcError")]
    assert len(lines) == 2 and all("RCCL could not be loaded" in ln and "libdoes_not_exist_bmc" in ln
                                   for ln in lines), r.stdout
