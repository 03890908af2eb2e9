#!/usr/bin/env bash
# round-4 GPU session 18: staging rows allocated out of the reserve (optional): GPU suite, the headline, rank rehearsals on one GPU on the
# final schedule (the N > 1 bench line)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s18_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s18_pytest.log
[ $rc -eq 0 ] || exit $rc
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/s18_$name.json 2> gpurun_out/s18_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/s18_$name.err; return 0; }
  python3 - "$name" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/s18_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), d["config"].get("n_chunks"), d["config"].get("pipe_steps"), d["config"].get("chained_groups"), d["config"].get("chained_groups_staged"), d.get("hbm_plan_gb"), d.get("kernel_ms_per_step"), (d.get("tile_solo") or {}).get("ms_per_step"), d.get("efficiency_vs_solo_tile"), d.get("rccl"))
PY
}
C="--steps 3 --warmup 1 --no-cpu-baseline"
run head --profile --steps 3 --warmup 1
run reh2_chunked --gpus 2 --tile-rows 1024 --tile-cols 512 --chunk 2192 --pipe 1104 $C
run reh4_chunked --gpus 4 --tile-rows 512 --tile-cols 512 --chunk 2192 --pipe 1104 $C --no-tile-solo
run reh6 --gpus 6 --tile-rows 1024 --tile-cols 176 --pipe 1104 $C --no-tile-solo
