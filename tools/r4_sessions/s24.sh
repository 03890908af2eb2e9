#!/usr/bin/env bash
# round-4 GPU session 24: rank rehearsals over RCCL on one GPU with the staging rows forced on / off: same cost to the last bit, timing
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/s24_$name.json 2> gpurun_out/s24_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/s24_$name.err; return 0; }
  python3 - "$name" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/s24_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), "cost", repr(d.get("cost")), d["config"].get("chained_groups"), d["config"].get("chained_groups_staged"), d.get("kernel_ms_per_step"), d.get("rccl"))
PY
}
C="--steps 3 --warmup 1 --no-cpu-baseline --no-tile-solo"
SMASHX_CHAIN_STAGE=1 run reh2c_stage1 --gpus 2 --tile-rows 1024 --tile-cols 512 --chunk 2192 --pipe 1104 $C
SMASHX_CHAIN_STAGE=0 run reh2c_stage0 --gpus 2 --tile-rows 1024 --tile-cols 512 --chunk 2192 --pipe 1104 $C
SMASHX_CHAIN_STAGE=1 run reh4c_stage1 --gpus 4 --tile-rows 512 --tile-cols 512 --chunk 2192 --pipe 1104 $C
SMASHX_CHAIN_STAGE=0 run reh4c_stage0 --gpus 4 --tile-rows 512 --tile-cols 512 --chunk 2192 --pipe 1104 $C
