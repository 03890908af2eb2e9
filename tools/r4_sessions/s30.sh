#!/usr/bin/env bash
# round-4 GPU session 30: rows requested ahead in the copy passes of the chained launches (SX_STG_PF 8 / 16 / 32: variants/lib_pf*.so)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 tools/ab_matrix.py --tag s30 --timeout 300 --steps 3 --warmup 1 -- \
  "pf16|base||" "pf32|pf32||" "pf8|pf8||" "pf16_b|base||" "pf32_b|pf32||" \
  "tile_pf16|base||--of 8 --as-rank 0" "tile_pf32|pf32||--of 8 --as-rank 0"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; rm -rf /tmp/st30
SMASHX_LIB=$GRAFT_REPO_ROOT/variants/lib_pf32.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st30 -- python3 bench.py --profile --steps 2 --warmup 1 > gpurun_out/s30_run.log 2>&1; echo "stats rc=$?"
f=$(find /tmp/st30 -name "*kernel_stats.csv" | head -1); grep -E "transpose" "$f" | cut -c1-150
