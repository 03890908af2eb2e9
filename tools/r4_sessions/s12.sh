#!/usr/bin/env bash
# round-4 GPU session 12: staging rows of the chained groups, fourth variant (LDS-FIFO transposition passes): suite, A/B on one box, kernel stats
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s12_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/s12_pytest.log
python3 tools/ab_matrix.py --tag s12 --timeout 300 --steps 3 --warmup 1 -- \
  "stage1|base||" \
  "stage0|base|SMASHX_CHAIN_STAGE=0|" \
  "stage1_b|base||" \
  "stage0_b|base|SMASHX_CHAIN_STAGE=0|" \
  "g1024_stage1|base||--grid 1024" \
  "g1024_stage0|base|SMASHX_CHAIN_STAGE=0|--grid 1024" \
  "tile_stage1|base||--of 8 --as-rank 0" \
  "tile_stage0|base|SMASHX_CHAIN_STAGE=0|--of 8 --as-rank 0" \
  "france_stage1|base||--mesh france:all" \
  "france_stage0|base|SMASHX_CHAIN_STAGE=0|--mesh france:all" \
  "fwd1024_stage1|base||--grid 1024 --forward-only" \
  "fwd1024_stage0|base|SMASHX_CHAIN_STAGE=0|--grid 1024 --forward-only"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; rm -rf /tmp/st12
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st12 -- python3 bench.py --profile --grid 1024 --steps 3 --warmup 1 > gpurun_out/s12_run.log 2>&1; echo "stats rc=$?"
f=$(find /tmp/st12 -name "*kernel_stats.csv" | head -1); head -1 "$f" > gpurun_out/s12_stats_1024.csv; grep -E "sx_k_" "$f" >> gpurun_out/s12_stats_1024.csv; grep -E "transpose|route" gpurun_out/s12_stats_1024.csv | cut -c1-160
