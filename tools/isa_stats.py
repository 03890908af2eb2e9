#!/usr/bin/env python3
"""Instruction mix of one kernel of libsmashx from the device assembly (hipcc -S --cuda-device-only):
    python tools/isa_stats.py <mangled-name-substring> [--loop]
Prints total instruction count and the counts per class (VALU fp32 / fp64 / transcendental / VMEM / SMEM / LDS / SALU / branches / waits)."""
import re
import subprocess
import sys
from collections import Counter

ASM = "/tmp/smashx_dev.s"


def build():
    import os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-ffp-contract=off", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S",
                           "-o", ASM, os.path.join(root, "smash_amd/csrc/smashx.hip")] + [a for a in sys.argv[2:] if a.startswith("-D")],
                          stderr=subprocess.DEVNULL)


def classify(op):
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")): return "valu_trans"
    if op.startswith("v_") and ("f64" in op): return "valu_f64"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_"): return "valu"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    if "--build" in sys.argv:
        build()
    s = open(ASM).read()
    key = sys.argv[1]
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", s, re.S | re.M):
        name = m.group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if key not in name and key not in dem:
            continue
        lines = [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
        c = Counter(classify(l.split()[0]) for l in lines)
        print(f"{dem[:70]:70s} total {len(lines):5d}  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()
