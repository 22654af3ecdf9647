#!/usr/bin/env python3
"""Build-time gate: no kernel of libpybmc_amd.so may spill.

Reads the `-Rpass-analysis=kernel-resource-usage` remarks hipcc wrote while compiling each
.hip (the Makefile keeps them as <source>.res) and fails when any kernel reports VGPR/SGPR
spills or scratch, unless its demangled name matches an entry of spill_whitelist.txt (which
must state the measured cost).  Usage: check_resources.py <file.res>...  [--list]
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def parse(path):
    out, cur = [], None
    pat = re.compile(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|"
                     r"SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s+(\S+)")
    for line in open(path, errors="replace"):
        m = pat.search(line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": val, "file": os.path.basename(path)}
            out.append(cur)
        elif cur is not None:
            cur[key.split(" [")[0]] = int(val)
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        d = r.stdout.splitlines()
        if len(d) == len(names):
            return d
    except OSError:
        pass
    return names


def whitelist():
    path = os.path.join(HERE, "spill_whitelist.txt")
    pats = []
    if os.path.exists(path):
        for line in open(path):
            line = line.split("#", 1)[0].strip()
            if line:
                pats.append(re.compile(line))
    return pats


def main(argv):
    files = [a for a in argv if not a.startswith("--")]
    kernels = [k for f in files for k in parse(f)]
    if not kernels:
        print("check_resources: no resource remarks found", file=sys.stderr)
        return 1
    for k, d in zip(kernels, demangle([k["name"] for k in kernels])):
        k["demangled"] = d
    wl = whitelist()
    bad = []
    for k in kernels:
        # SGPR spills live in VGPR lanes (v_writelane/v_readlane), not in memory: reported by
        # --list, not a failure.  VGPR spills and scratch are memory traffic inside the kernel.
        if k.get("VGPRs Spill", 0) or k.get("ScratchSize", 0):
            k["whitelisted"] = any(p.search(k["demangled"]) for p in wl)
            if not k["whitelisted"]:
                bad.append(k)
    if "--list" in argv:
        for k in kernels:
            print(f'{k.get("VGPRs", 0):4d} vgpr {k.get("AGPRs", 0):3d} agpr {k.get("TotalSGPRs", 0):4d} sgpr '
                  f'{k.get("VGPRs Spill", 0):3d} vspill {k.get("ScratchSize", 0):4d} B scratch  {k["demangled"]}')
    for k in kernels:
        if k.get("whitelisted"):
            print(f'whitelisted: {k["demangled"]}: {k.get("VGPRs Spill", 0)} VGPR spills, '
                  f'{k.get("ScratchSize", 0)} B/lane scratch')
    for k in bad:
        print(f'SPILL: {k["demangled"]}: {k.get("VGPRs Spill", 0)} VGPR spills, '
              f'{k.get("SGPRs Spill", 0)} SGPR spills, {k.get("ScratchSize", 0)} B/lane scratch '
              f'({k["file"]})', file=sys.stderr)
    print(f"check_resources: {len(kernels)} kernels, {len(bad)} spilling (not whitelisted)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
