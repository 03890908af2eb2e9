#!/usr/bin/env bash
# Where the vertical kernels' cycles go, by instruction class (1024^2 x 8760, one dispatch each):  tools/pmc_issue_probe.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf /tmp/pi
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU \
    --output-format csv -d /tmp/pi -- python3 bench.py --profile --grid 1024 --steps 1 --warmup 0 > gpurun_out/pi.log 2>&1 || { echo failed; tail -5 gpurun_out/pi.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pi/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "sx_k_" in n: acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n in sorted(acc):
    a = acc[n]; wc = a["SQ_WAVE_CYCLES"] or 1.0
    print(n, {k: round(v / wc, 4) for k, v in a.items() if k != "SQ_WAVE_CYCLES"}, "wave_cycles %.3g busy %.3g" % (wc, a["SQ_BUSY_CYCLES"]))
PY
