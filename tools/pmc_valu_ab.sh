#!/usr/bin/env bash
# Executed instruction counts of the vertical kernels for several library builds, on the 1024^2 x 8760 store-all case (one dispatch
# = 1024^2 x 8760 cell-steps):   tools/pmc_valu_ab.sh name ...      ("base" = smash_amd/libsmashx.so, else variants/lib_<name>.so)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for k in "$@"; do
  lib=$PWD/variants/lib_$k.so; [ "$k" = base ] && lib=$PWD/smash_amd/libsmashx.so
  export SMASHX_LIB=$lib
  rm -rf /tmp/pv_$k
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU \
      --output-format csv -d /tmp/pv_$k -- python3 bench.py --profile --grid 1024 --steps 1 --warmup 0 ${PV_ARGS:-} > gpurun_out/pv_$k.log 2>&1 || { echo "$k failed"; tail -3 gpurun_out/pv_$k.log; exit 1; }
  python3 - "$k" <<'PY'
import csv, glob, sys, collections
k = sys.argv[1]
f = glob.glob(f"/tmp/pv_{k}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "sx_k_vert" not in n: continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": nd[n] += 1
cs = 1024 * 1024 * 8760 / 64.0
for n in sorted(acc):
    a = acc[n]; wc = a["SQ_WAVE_CYCLES"]
    print(k, n, "dispatches", nd[n], "valu/cs %.1f salu/cs %.1f" % (a["SQ_INSTS_VALU"] / cs / nd[n], a["SQ_INSTS_SALU"] / cs / nd[n]),
          "issuing %.3f waitcnt %.3f issue_stalled %.3f" % (a["SQ_ACTIVE_INST_ANY"] / wc, a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc))
PY
done
