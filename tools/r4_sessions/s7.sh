#!/usr/bin/env bash
# round-4 GPU session 7: suite on the cleaned-up library, the driver's bench command, profiles of the headline and of the forward-only case
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s7_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/s7_pytest.log
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step","kernel_ms_per_step","hbm_plan_gb","hbm_free_at_plan_gb")})
for k in ("secondary","tile_solo","forward_only","real_d8","exact_libm"):
    o=d.get(k,{}); print(k, {a:o.get(a) for a in ("value","ms_per_step","error","slowdown_vs_default")}, (o.get("headline") or {}).get("ms_per_step"), (o.get("synthetic_d8_1024") or {}).get("ms_per_step"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("solo_core",{}).get("value"))
PY
bash tools/profile_round.sh r4 --steps 3 --warmup 1 && echo "profile 2048 ok"
bash tools/profile_round.sh r4fwd --steps 3 --warmup 1 --grid 1024 --forward-only && echo "profile fwd ok"
ls gpurun_out | grep -E "^r4" | head -30
