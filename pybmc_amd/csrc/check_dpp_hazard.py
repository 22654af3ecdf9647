#!/usr/bin/env python3
"""Build-time gate: no inline-asm DPP FMA may read a source a VALU instruction has just written.

``v_fmac_f64_dpp acc, u_rows, -x row_newbcast:n`` (bmc_loop.h, fmac_rowbcast_neg) is emitted by
inline asm, which LLVM's hazard recognizer does not look into.  gfx950 needs 2 wait states
between a VALU write of a VGPR and a DPP read of it, and 5 between a VALU write of EXEC
(v_cmpx) and any DPP instruction (cdna guide 5.7 item 2).  The asm opens every 16-column block
with ``s_nop 1``; inside a block nothing in the source text writes u_rows -- but a register
copy, reload or re-materialisation placed there by the register allocator would, silently, in
any of the ~270 kernel instantiations.  This script disassembles the gfx950 code objects of the
given host objects and fails the build if a straight-line VALU write of the DPP source (or of
EXEC) sits inside the required window.  Usage: check_dpp_hazard.py <file.o>...
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("BMC_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
REG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def regs(op):
    m = REG.match(op.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def split_ops(rest):
    return [o.strip() for o in rest.split("//")[0].split(",")]


def scan(lines):
    """lines: disassembly text.  Returns (number of DPP FMAs, list of hazard descriptions)."""
    hist = []          # (mnemonic, operands, wait states it provides, text) of the current run
    n_dpp, bad = 0, []
    func = "?"
    for raw in lines:
        line = raw.rstrip("\n")
        if not line.startswith("\t") and not line.startswith(" "):
            s = line.strip()
            if s.endswith(":"):      # a label: the predecessors beyond it are unknown
                m = re.search(r"<([^>]+)>:$", s)
                if m and not m.group(1).startswith("L"):
                    func = m.group(1)
                hist = []
            continue
        body = line.strip()
        if not body or body.startswith("//"):
            continue
        parts = body.split(None, 1)
        mn = parts[0]
        ops = split_ops(parts[1]) if len(parts) > 1 else []
        text = body.split("//")[0].strip()
        if mn.startswith("v_fmac_f64_dpp"):
            n_dpp += 1
            src = regs(ops[1]) if len(ops) > 1 else set()
            waited = 0
            for pm, pops, ws, ptxt in reversed(hist):
                if waited >= 5:
                    break
                if pm.startswith("v_"):
                    if pm.startswith("v_cmpx"):
                        bad.append(f"{func}: {ptxt}  ->  {text}  (EXEC written {waited} wait "
                                   f"states before a DPP read, 5 needed)")
                    elif waited < 2 and pops and (regs(pops[0]) & src):
                        bad.append(f"{func}: {ptxt}  ->  {text}  (DPP source written {waited} "
                                   f"wait states before the read, 2 needed)")
                waited += ws
        ws = 1
        if mn == "s_nop" and ops:
            try:
                ws = int(ops[0], 0) + 1
            except ValueError:
                ws = 1
        if mn.startswith("s_cbranch") or mn.startswith("s_branch") or mn == "s_endpgm":
            hist = []      # (a taken branch lands on a label; the fall-through starts afresh)
            continue
        hist.append((mn, ops, ws, text))
        if len(hist) > 8:
            hist.pop(0)
    return n_dpp, bad


def device_asm(obj, tmp):
    # llvm-objdump --offloading writes <input>.<n>.hipv4-amdgcn-amd-amdhsa--gfx950 next to its
    # input: work on a copy in the scratch directory
    import glob
    import shutil
    copy = os.path.join(tmp, os.path.basename(obj))
    shutil.copyfile(obj, copy)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True,
                   capture_output=True)
    out = []
    for co in sorted(glob.glob(copy + ".*gfx950*")):
        r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co],
                           check=True, capture_output=True, text=True)
        out += r.stdout.splitlines()
    if not out:
        raise SystemExit(f"check_dpp_hazard: no gfx950 code object found in {obj}")
    return out


def main(argv):
    total, bad = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in argv:
            n, b = scan(device_asm(obj, tmp))
            total += n
            bad += [f"{os.path.basename(obj)}: {x}" for x in b]
    if bad:
        print(f"check_dpp_hazard: {len(bad)} DPP hazard(s):", file=sys.stderr)
        for x in bad[:40]:
            print("  " + x, file=sys.stderr)
        return 1
    print(f"check_dpp_hazard: {total} inline-asm DPP FMAs, none inside a VALU-write window")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
