#!/usr/bin/env bash
# round-4 GPU session 27 (final tree: inlet-order schedule, staging rows, early release of the chunk buffers): suite, the driver's bench command, profiles of the headline
# and of the forward-only case, parity tables of both builds
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s27_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s27_pytest.log
[ $rc -eq 0 ] || exit $rc
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4c_bench_default.json 2> gpurun_out/r4c_bench_default.err; rc=$?; echo "bench rc=$rc in $(( $(date +%s) - t0 )) s"
[ $rc -eq 0 ] || exit $rc
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4c_bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step","kernel_ms_per_step","hbm_plan_gb","hbm_free_at_plan_gb")})
print(d["config"].get("chained_groups"), d["config"].get("chained_groups_staged"), d.get("chained_launch_ms_per_step"))
for k in ("secondary","tile_solo","forward_only","real_d8","exact_libm"):
    o=d.get(k,{}); print(k, {a:o.get(a) for a in ("value","ms_per_step","error","slowdown_vs_default","chained_groups_staged")}, (o.get("headline") or {}).get("ms_per_step"), (o.get("synthetic_d8_1024") or {}).get("ms_per_step"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("solo_core",{}).get("value"))
print("roofline", d.get("roofline"))
PY
bash tools/profile_round.sh r4c --steps 3 --warmup 1 && echo "profile 2048 ok" || exit 1
bash tools/profile_round.sh r4cfwd --steps 3 --warmup 1 --grid 1024 --forward-only && echo "profile fwd ok" || exit 1
python3 tools/parity_table.py > gpurun_out/r4c_parity_default.md 2> gpurun_out/s27_pt_default.log; echo "parity default rc=$?"
SMASHX_EXACT_LIBM=1 python3 tools/parity_table.py --assert-exact > gpurun_out/r4c_parity_exact.md 2> gpurun_out/s27_pt_exact.log; echo "parity exact rc=$?"
tail -3 gpurun_out/r4c_parity_default.md | cut -c1-300; tail -3 gpurun_out/r4c_parity_exact.md | cut -c1-300
ls gpurun_out | grep -E "^r4c" | head -30
