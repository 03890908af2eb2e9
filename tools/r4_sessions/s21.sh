#!/usr/bin/env bash
# round-4 GPU session 21: inputs of the pipeline model on the final schedule: solo tile at three sub-chunk lengths, store-all rehearsals
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/s21_$name.json 2> gpurun_out/s21_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/s21_$name.err; return 0; }
  python3 - "$name" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/s21_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), d["config"].get("n_chunks"), d["config"].get("pipe_steps"), d["config"].get("chained_groups_staged"), d.get("kernel_ms_per_step"), d.get("chained_launch_ms_per_step"))
PY
}
C="--steps 3 --warmup 1 --no-cpu-baseline"
run solo_p4384 --of 8 --as-rank 0 --pipe 4384 $C
run solo_p2192 --of 8 --as-rank 0 --pipe 2192 $C
run solo_p1104 --of 8 --as-rank 0 --pipe 1104 $C
run reh2 --gpus 2 --tile-rows 1024 --tile-cols 512 $C --no-tile-solo
run reh4 --gpus 4 --tile-rows 512 --tile-cols 512 $C --no-tile-solo
