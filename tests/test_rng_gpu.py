"""The on-device generator (-m gpu): Philox4x32-10 bit-for-bit against a plain
Python restatement of the published algorithm and its known-answer vectors, then
distributional checks of the N(0,1) and Gamma(a,1) transforms."""
import numpy as np
import pytest
from scipy import stats

from gpu_common import gpu_ctx

pytestmark = pytest.mark.gpu

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k[1]) & MASK, p0 & MASK]
        k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
    return c


def test_python_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox4x32_10([MASK] * 4, [MASK] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                         [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_device_philox_bit_exact():
    ctx = gpu_ctx()
    for seed, stream in [(0, 0), (0x299f31d0a4093822, 0x13198a2e), (2 ** 64 - 1, 0xFFFFFFFF)]:
        got = ctx.philox_raw(seed, stream, 300)
        for i in (0, 1, 2, 63, 64, 255, 299):
            want = philox4x32_10([i, 0, stream, 0], [seed & MASK, seed >> 32])
            assert got[i].tolist() == want


def test_normals():
    ctx = gpu_ctx()
    z, _ = ctx.rng_fill(12345, n_normal=2_000_001)
    assert abs(z.mean()) < 5 / np.sqrt(len(z))
    assert abs(z.var() - 1) < 5 * np.sqrt(2 / len(z))
    assert abs(stats.skew(z)) < 0.01 and abs(stats.kurtosis(z)) < 0.02
    assert stats.kstest(z[:200000], "norm").pvalue > 1e-4
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5 / np.sqrt(len(z))
    z2, _ = ctx.rng_fill(12345, n_normal=1000)
    assert np.array_equal(z2, z[:1000])            # counter-based: prefix-stable
    z3, _ = ctx.rng_fill(12346, n_normal=1000)
    assert not np.array_equal(z3, z2)


@pytest.mark.parametrize("shape", [0.5, 0.9, 1.0, 2.5, 75.5, 5000.5, 100000.5])
def test_gammas(shape):
    ctx = gpu_ctx()
    n = 400000
    _, g = ctx.rng_fill(99, shape=shape, n_gamma=n)
    assert np.all(g > 0) and np.isfinite(g).all()
    assert abs(g.mean() - shape) < 5 * np.sqrt(shape / n)
    assert abs(g.var() / shape - 1) < 0.02
    assert stats.kstest(g[:100000], "gamma", args=(shape,)).pvalue > 1e-4
