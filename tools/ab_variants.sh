#!/usr/bin/env bash
# A/B of library builds on ONE box (box-to-box spread is +-3 %):  tools/ab_variants.sh <bench args> -- name[=path] ...
# "base" = smash_amd/libsmashx.so; other names = variants/lib_<name>.so.  Each runs `python3 bench.py --profile <bench args>`.
set -u
cd "$(dirname "$0")/.."
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
mkdir -p gpurun_out
for k in "$@"; do
  lib=$PWD/variants/lib_$k.so; [ "$k" = base ] && lib=$PWD/smash_amd/libsmashx.so
  SMASHX_LIB=$lib timeout -k 10 400 python3 bench.py --profile "${args[@]}" > gpurun_out/ab_$k.json 2> gpurun_out/ab_$k.err || { echo "$k failed"; tail -3 gpurun_out/ab_$k.err; continue; }
  python3 - "$k" <<'PY'
import json, sys
k = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab_{k}.json").read().strip().splitlines()[-1])
print(k, round(d["ms_per_step"], 2), d.get("kernel_ms_per_step"))
PY
done
