#!/usr/bin/env bash
# round-4 GPU session 34: the round's last tree (four wave-blocks per copy workgroup in): GPU suite, smoke, the driver's bench command
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s34_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s34_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s34_bench_default.json 2> gpurun_out/s34_bench_default.err; rc=$?; echo "bench rc=$rc in $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/s34_bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step","kernel_ms_per_step","hbm_plan_gb","hbm_free_at_plan_gb")})
print(d["config"].get("n_chunks"), d["config"].get("chained_groups"), d["config"].get("chained_groups_staged"), d.get("chained_launch_ms_per_step"))
r=d["roofline"]; print({k:r[k] for k in ("achieved","frac","traffic","avg_launch_ms")}, r["valu"]["instr_per_cellstep"], r["valu"]["weighted"]["frac"])
for k in ("secondary","tile_solo","forward_only","real_d8","exact_libm"):
    o=d.get(k,{}); print(k, {a:o.get(a) for a in ("value","ms_per_step","error","slowdown_vs_default")}, (o.get("headline") or {}).get("ms_per_step"))
PY
