"""The build-time gates of pybmc_amd/csrc (CPU): the DPP-hazard scanner must see what it is
there to see, and the RCCL loader must fail with a status, not a crash, when RCCL is absent."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_scanner():
    spec = importlib.util.spec_from_file_location(
        "check_dpp_hazard", os.path.join(ROOT, "pybmc_amd", "csrc", "check_dpp_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


DPP = "\tv_fmac_f64_dpp v[38:39], v[36:37], -v[40:41] row_newbcast:%d row_mask:0xf bank_mask:0xf"


def test_dpp_hazard_scanner_flags_a_write_inside_the_window():
    m = load_scanner()
    clean = ["0000 <kernel_a>:", "\tds_read_b64 v[36:37], v1", "\ts_waitcnt lgkmcnt(0)", "\ts_nop 1",
             DPP % 0, "\tv_cvt_f64_f32_e32 v[40:41], v29", DPP % 1]
    n, bad = m.scan(clean)
    assert n == 2 and bad == []
    # a copy of the DPP source right in front of a statement that has no s_nop of its own
    hazard = clean[:5] + ["\tv_mov_b32_e32 v37, v3", DPP % 1]
    n, bad = m.scan(hazard)
    assert n == 2 and len(bad) == 1 and "kernel_a" in bad[0] and "2 needed" in bad[0]
    # one independent instruction in between is one wait state: still too close
    n, bad = m.scan(clean[:5] + ["\tv_mov_b32_e32 v36, v3", "\tv_add_u32_e32 v9, v9, v9", DPP % 1])
    assert len(bad) == 1
    # two wait states: fine
    n, bad = m.scan(clean[:5] + ["\tv_mov_b32_e32 v36, v3", "\ts_nop 1", DPP % 1])
    assert bad == []
    # a VALU write of EXEC needs five
    n, bad = m.scan(clean[:5] + ["\tv_cmpx_lt_f64_e32 v[2:3], v[4:5]", "\ts_nop 2", DPP % 1])
    assert len(bad) == 1 and "5 needed" in bad[0]
    n, bad = m.scan(clean[:5] + ["\tv_cmpx_lt_f64_e32 v[2:3], v[4:5]", "\ts_nop 4", DPP % 1])
    assert bad == []


def test_dpp_hazard_gate_passes_on_the_built_objects():
    obj = os.path.join(ROOT, "pybmc_amd", "csrc", "kernels_gibbs.o")
    if not os.path.exists(obj):      # (objects are build products; build() makes them)
        import pytest
        pytest.skip("kernels_gibbs.o not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "pybmc_amd", "csrc", "check_dpp_hazard.py"),
                        obj], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "inline-asm DPP FMAs" in r.stdout


def test_rccl_load_failure_is_a_status_not_a_crash():
    """bmc_comm_unique_id with an RCCL that cannot be loaded (BMC_RCCL_SONAME names a library
    that does not exist) returns BMC_EHIP.  Round-2 advisor finding: the failure path called
    dlerror() twice and built a std::string from NULL."""
    code = ("import ctypes, sys; sys.path.insert(0, %r)\n"
            "from pybmc_amd import _lib\n"
            "lib = _lib.load_library()\n"
            "buf = ctypes.create_string_buffer(128)\n"
            "rc1 = lib.bmc_comm_unique_id(buf); rc2 = lib.bmc_comm_unique_id(buf)\n"
            "print('rc', rc1, rc2)\n" % ROOT)
    env = dict(os.environ, BMC_RCCL_SONAME="libdoes_not_exist_bmc.so.9")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rc 3 3" in r.stdout          # BMC_EHIP twice (the second call sees the cached state)
