#!/usr/bin/env bash
# round-4 GPU session 15: roots below the chain write their series straight into the staging rows (x_stg): whole GPU suite, then A/B
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/s15_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s15_pytest.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_matrix.py --tag s15 --timeout 300 --steps 3 --warmup 1 -- \
  "direct1|base||" \
  "direct0|base|SMASHX_STAGE_DIRECT=0|" \
  "stage0|base|SMASHX_CHAIN_STAGE=0|" \
  "direct1_b|base||" \
  "direct0_b|base|SMASHX_STAGE_DIRECT=0|" \
  "g1024_direct1|base||--grid 1024" \
  "g1024_direct0|base|SMASHX_STAGE_DIRECT=0|--grid 1024" \
  "g1024_stage0|base|SMASHX_CHAIN_STAGE=0|--grid 1024" \
  "tile_direct1|base||--of 8 --as-rank 0" \
  "tile_direct0|base|SMASHX_STAGE_DIRECT=0|--of 8 --as-rank 0"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; rm -rf /tmp/st15
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st15 -- python3 bench.py --profile --steps 2 --warmup 1 > gpurun_out/s15_run.log 2>&1; echo "stats rc=$?"
f=$(find /tmp/st15 -name "*kernel_stats.csv" | head -1); head -1 "$f" > gpurun_out/s15_stats_2048.csv; grep -E "sx_k_" "$f" >> gpurun_out/s15_stats_2048.csv; cut -c1-170 gpurun_out/s15_stats_2048.csv
