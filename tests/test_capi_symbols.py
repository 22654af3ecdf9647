"""CPU-side checks of the drop-in boundary: the shared library loads and exports
every symbol include/pybmc_amd.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from pybmc_amd import _lib

HEADER = os.path.join(ROOT, "include", "pybmc_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bmc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_symbols()
    for must in ("bmc_create", "bmc_destroy", "bmc_set_problem", "bmc_set_prior",
                 "bmc_gibbs_run", "bmc_gibbs_run_device", "bmc_residual_rss", "bmc_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build the library first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_python_binding_covers_the_header():
    assert sorted(_lib.PROTOTYPES) == declared_symbols()
    lib = _lib.load_library()
    assert lib.bmc_abi_version() == _lib.ABI_VERSION == 4


def test_struct_layouts_match_the_header():
    # bmc_stats: 4 doubles, int64, 7 int32 (+4 pad), 2 int64 -> 8-byte aligned
    assert ctypes.sizeof(_lib.Stats) == 4 * 8 + 8 + 8 * 4 + 2 * 8
    assert ctypes.sizeof(_lib.Tuning) == 32


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpybmc_amd.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load_library()


def test_no_gpu_means_no_context():
    """On a box without a gfx950 device creating a context must raise (never fall back)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.BmcError, match="no CPU fallback"):
        _lib.Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pybmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_graft_entry_build_passes():
    """The driver's build check: make (a no-op when the library is current), import, ABI version."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    entry = importlib.import_module("__graft_entry__")
    entry.build()


def test_header_is_plain_c(tmp_path):
    """include/pybmc_amd.h is the drop-in boundary: it must compile as C99 on its own (no C++,
    no HIP or torch types), and a C caller must link against the library by name alone."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "pybmc_amd.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = tmp_path / "caller.c"
    src.write_text(
        '#include "pybmc_amd.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "    bmc_ctx* ctx = 0;\n"
        "    printf(\"abi %d\\n\", bmc_abi_version());\n"
        "    /* no GPU in the build container: create must fail with a status, not crash */\n"
        "    int rc = bmc_create(0, &ctx);\n"
        "    printf(\"create %d\\n\", rc);\n"
        "    if (rc == BMC_OK) bmc_destroy(ctx);\n"
        "    return 0;\n}\n")
    exe = tmp_path / "caller"
    libdir = os.path.join(ROOT, "pybmc_amd")
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", libdir, "-lpybmc_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "abi 4" in r.stdout and "create" in r.stdout
