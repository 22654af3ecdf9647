"""pybmc_amd/csrc/bmc_math.h on the CPU: the elementary functions of the on-device normal
generator (Box-Muller: log on (0, 1], sin / cos of 2 pi u) are plain C++ that the gfx950 kernels
and this test compile from the same text; here g++ builds tests/host_math_check.cpp and the
worst errors against long-double libm must stay within 1-2 ulp."""
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_generator_math_against_libm(tmp_path):
    exe = tmp_path / "host_math_check"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-o", str(exe),
                    os.path.join(HERE, "host_math_check.cpp")], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    wl, ws, wc = (float(v) for v in out[0].split())
    assert wl <= 1.01 and ws <= 2.01 and wc <= 2.01, (wl, ws, wc)
    # quadrant boundaries are exact
    want = [(0.0, 1.0), (1.0, 0.0), (0.0, -1.0), (-1.0, 0.0), (0.0, 1.0)]
    for line, (s, c) in zip(out[1:6], want):
        sn, cs = (float(v) for v in line.split())
        assert sn == s and cs == c
    l1, lmin = (float(v) for v in out[6].split())
    assert l1 == 0.0 and abs(lmin - math.log(2.0 ** -53)) <= 1e-15 * abs(lmin)
    z0, z1 = (float(v) for v in out[7].split())
    rad = math.sqrt(-2.0 * math.log(0.5))
    assert np.allclose([z0, z1], [rad * math.cos(2 * math.pi * 0.125), rad * math.sin(2 * math.pi * 0.125)],
                       rtol=1e-15, atol=0)
