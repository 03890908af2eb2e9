#!/usr/bin/env bash
# A/B of library builds on rank 0's tile of the 8-rank decomposition, alone on the GPU:  tools/ab_tile.sh name ...   (see ab_variants.sh)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for k in "$@"; do
  lib=$PWD/variants/lib_$k.so; [ "$k" = base ] && lib=$PWD/smash_amd/libsmashx.so
  SMASHX_LIB=$lib timeout -k 10 300 python3 bench.py --of 8 --as-rank 0 --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/abt_$k.err | tail -1 > gpurun_out/abt_$k.json || { echo "$k failed"; continue; }
  python3 - "$k" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/abt_{sys.argv[1]}.json"))
print(sys.argv[1], round(d["ms_per_step"], 2), d.get("kernel_ms_per_step"))
PY
done
