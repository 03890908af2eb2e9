#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (counter_collection.csv) per kernel name: sum of each counter over dispatches and the
number of dispatches.  With --json OUT --cellsteps N (cell-steps one whole-period launch processes) also writes the per-kernel
figures bench.py attaches to its roofline object: HBM bytes per cell-step (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, both in KiB:
MI355X_MICROARCH.md "HBM") and VALU wave-instructions per cell-step (SQ_INSTS_VALU / wave-steps)."""
import argparse
import collections
import csv
import glob
import json
import os

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("--json", default="")
ap.add_argument("--cellsteps", type=float, default=0.0)
ap.add_argument("--command", default="")
a = ap.parse_args()
acc = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(a.root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "sx_k" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} {v:.6g}   ({len(ndisp[k][c])} dispatches)")
if a.json and a.cellsteps > 0:
    out = {}
    groups = {"sx_k_vert_fwd": lambda k: k.startswith("sx_k_vert_fwd<") and ", true," in k,       # the taped forward pass
              "sx_k_vert_adj": lambda k: k.startswith("sx_k_vert_adj<"),
              "sx_k_route_fwd": lambda k: k.startswith("sx_k_route_fwd<true"),
              "sx_k_route_adj": lambda k: k.startswith("sx_k_route_adj<")}
    for name, sel in groups.items():
        ks = [k for k in acc if sel(k)]
        if not ks:
            continue
        passes = max(len(ndisp[k]["FETCH_SIZE"]) for k in ks) if all("FETCH_SIZE" in acc[k] for k in ks) else 0
        if not passes:
            continue
        fetch = sum(acc[k]["FETCH_SIZE"] for k in ks) / passes
        write = sum(acc[k].get("WRITE_SIZE", 0.0) for k in ks) / max(max(len(ndisp[k]["WRITE_SIZE"]) for k in ks), 1)
        e = {"kernel": " + ".join(ks), "fetch_size_kib_per_pass": fetch, "write_size_kib_per_pass": write, "cellsteps_per_pass": a.cellsteps,
             "hbm_bytes_per_cellstep_corrected": (2.0 * fetch + write) * 1024.0 / a.cellsteps,
             "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md 'HBM'); WRITE_SIZE exact; a routing pass = its round-0 launch + the chained launch",
             "command": a.command}
        if all("SQ_INSTS_VALU" in acc[k] for k in ks):
            pv = max(len(ndisp[k]["SQ_INSTS_VALU"]) for k in ks)
            e["valu_per_cellstep"] = sum(acc[k]["SQ_INSTS_VALU"] for k in ks) / pv / (a.cellsteps / 64.0)
            e["salu_per_cellstep"] = sum(acc[k].get("SQ_INSTS_SALU", 0.0) for k in ks) / pv / (a.cellsteps / 64.0)
            wc = sum(acc[k].get("SQ_WAVE_CYCLES", 0.0) for k in ks)
            if wc > 0:
                e["wave_cycles_share"] = {"issuing": sum(acc[k].get("SQ_ACTIVE_INST_ANY", 0.0) for k in ks) / wc,
                                          "parked_on_waitcnt": sum(acc[k].get("SQ_WAIT_ANY", 0.0) for k in ks) / wc,
                                          "issue_stalled": sum(acc[k].get("SQ_WAIT_INST_ANY", 0.0) for k in ks) / wc}
        out[name] = e
    json.dump(out, open(a.json, "w"), indent=1)
