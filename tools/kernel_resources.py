#!/usr/bin/env python3
"""VGPRs / occupancy / spills of every kernel in libsmashx (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
Extra arguments go to hipcc (e.g. -DSX_EXACT_LIBM=1)."""
import os
import re
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-ffp-contract=off", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-o", "/tmp/libsmashx_res.so",
       os.path.join(root, "smash_amd/csrc/smashx.hip"), os.path.join(root, "smash_amd/csrc/sx_plan.cpp"), "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for l in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", l)
    if not m:
        if "error" in l:
            print(l)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.split("\n")
for n, (k, r) in zip(names, rows.items()):
    n = re.sub(r"\(.*", "", n)
    g = lambda key: r.get(key, "?")
    print(f"{n:48s} VGPR {g('VGPRs'):>4s} AGPR {g('AGPRs'):>3s} occ {g('Occupancy [waves/SIMD]'):>2s} spillV {g('VGPRs Spill'):>3s} "
          f"scratch {g('ScratchSize [bytes/lane]'):>4s} LDS {g('LDS Size [bytes/block]')}")
