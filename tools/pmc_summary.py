#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (counter_collection.csv) per kernel family.

    pmc_summary.py <root> --bench <bench JSON of the same command, with profile_accounting> --json OUT

Every pass directory <root>/pmcN must hold the counters of exactly ONE process (tools/profile_round.sh runs `bench.py --profile`,
which starts no child); more than one is an error, not an average.  A family's counters are summed over ALL its dispatches in the
process and divided by the cell-steps those dispatches processed, which bench.py reports (profile_accounting: sweeps run, and the
cell-steps per sweep of the taped forward pass, the untaped forward pass and the reverse pass -- a checkpointed adjoint sweeps every
storage chunk but the last twice).  Per family: HBM bytes per cell-step (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, both in KiB:
MI355X_MICROARCH.md "HBM") and VALU wave-instructions per cell-step (SQ_INSTS_VALU / (cell-steps / 64))."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("--bench", required=True)
ap.add_argument("--json", default="")
ap.add_argument("--command", default="")
a = ap.parse_args()

bench = json.load(open(a.bench))
acct = bench["profile_accounting"]
nadj, nfwd = acct["adjoint_sweeps"], acct["forward_sweeps"]
pa, pf = acct["per_adjoint_sweep"], acct["per_forward_sweep"]
total = {"taped_forward": nadj * pa["taped_forward"],
         "untaped_forward": nadj * pa["untaped_forward"] + nfwd * pf["untaped_forward"],
         "reverse": nadj * pa["reverse"]}

acc = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(lambda: collections.defaultdict(int))
for pdir in sorted(glob.glob(os.path.join(a.root, "pmc*"))):
    if not os.path.isdir(pdir):
        continue
    files = glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True)
    if len(files) != 1:
        sys.exit(f"{pdir}: {len(files)} counter files -- expected the one process of `bench.py --profile` (a child process or launcher "
                 "ran under the profiler?)")
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "sx_k" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} {v:.6g}   ({ndisp[k][c]} dispatches)")


def tmpl(k):        # template arguments of a kernel name
    return [x.strip() for x in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k else []


families = {   # name -> (selector, cell-steps the family's dispatches processed in the whole process)
    "sx_k_vert_fwd": (lambda k: k.startswith("sx_k_vert_fwd<") and tmpl(k)[1] == "true", total["taped_forward"]),
    "sx_k_vert_fwd_untaped": (lambda k: k.startswith("sx_k_vert_fwd<") and tmpl(k)[1] == "false", total["untaped_forward"]),
    "sx_k_vert_adj": (lambda k: k.startswith("sx_k_vert_adj<"), total["reverse"]),
    "sx_k_route_fwd": (lambda k: k.startswith("sx_k_route_fwd<true"), total["taped_forward"]),
    "sx_k_route_fwd_untaped": (lambda k: k.startswith("sx_k_route_fwd<false"), total["untaped_forward"]),
    "sx_k_route_adj": (lambda k: k.startswith("sx_k_route_adj<"), total["reverse"]),
    # the copy passes of the chained launches (staging rows): per cell-step of the WHOLE domain, like the routing families they belong to
    "sx_k_chain_transpose_gather": (lambda k: k.startswith("sx_k_chain_transpose<true"), total["taped_forward"] + total["untaped_forward"]),
    "sx_k_chain_transpose_scatter": (lambda k: k.startswith("sx_k_chain_transpose<false"), total["reverse"]),
}
out = {"workload": {"grid": acct["grid"], "n_chunks": acct["n_chunks"], "chunk_steps": acct["chunk_steps"], "cellsteps_per_sweep": acct["cellsteps"],
                    "adjoint_sweeps_profiled": nadj, "forward_sweeps_profiled": nfwd, "forward_only": nadj == 0, "config": bench.get("config", {}).get("workload")}}
for name, (sel, cs) in families.items():
    ks = [k for k in acc if sel(k)]
    if not ks or cs <= 0 or not all("FETCH_SIZE" in acc[k] for k in ks):
        continue
    fetch = sum(acc[k]["FETCH_SIZE"] for k in ks)
    write = sum(acc[k].get("WRITE_SIZE", 0.0) for k in ks)
    e = {"kernel": " + ".join(ks), "dispatches": int(sum(ndisp[k]["FETCH_SIZE"] for k in ks)), "cellsteps_counted": cs,
         "fetch_size_kib": fetch, "write_size_kib": write,
         "hbm_bytes_per_cellstep_corrected": (2.0 * fetch + write) * 1024.0 / cs,
         "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md 'HBM'); WRITE_SIZE exact; counters summed over "
                       "every dispatch of the family in the profiled process / the cell-steps those dispatches swept (bench.py profile_accounting)",
         "command": a.command}
    if all("SQ_INSTS_VALU" in acc[k] for k in ks):
        e["valu_per_cellstep"] = sum(acc[k]["SQ_INSTS_VALU"] for k in ks) / (cs / 64.0)
        e["salu_per_cellstep"] = sum(acc[k].get("SQ_INSTS_SALU", 0.0) for k in ks) / (cs / 64.0)
        wc = sum(acc[k].get("SQ_WAVE_CYCLES", 0.0) for k in ks)
        if wc > 0:
            e["wave_cycles_share"] = {"issuing": sum(acc[k].get("SQ_ACTIVE_INST_ANY", 0.0) for k in ks) / wc,
                                      "parked_on_waitcnt": sum(acc[k].get("SQ_WAIT_ANY", 0.0) for k in ks) / wc,
                                      "issue_stalled": sum(acc[k].get("SQ_WAIT_INST_ANY", 0.0) for k in ks) / wc}
    if all("SQ_INSTS_VALU_FMA_F64" in acc[k] for k in ks):       # optional pass: the executed mix for the weighted VALU ceiling
        per = lambda c: sum(acc[k].get(c, 0.0) for k in ks) / (cs / 64.0)
        e["valu_mix_per_cellstep"] = {"f64": per("SQ_INSTS_VALU_ADD_F64") + per("SQ_INSTS_VALU_MUL_F64") + per("SQ_INSTS_VALU_FMA_F64"),
                                      "trans": per("SQ_INSTS_VALU_TRANS_F32") + per("SQ_INSTS_VALU_TRANS_F64"),
                                      "cvt": per("SQ_INSTS_VALU_CVT"), "int32": per("SQ_INSTS_VALU_INT32")}
    out[name] = e
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
