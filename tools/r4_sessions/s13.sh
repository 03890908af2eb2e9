#!/usr/bin/env bash
# round-4 GPU session 13: staging rows, fourth variant with a deeper prefetch in the transposition passes: A/B on one box, kernel stats
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "staging or chunking" > gpurun_out/s13_pytest.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/s13_pytest.log
python3 tools/ab_matrix.py --tag s13 --timeout 300 --steps 3 --warmup 1 -- \
  "stage1|base||" \
  "stage0|base|SMASHX_CHAIN_STAGE=0|" \
  "stage1_b|base||" \
  "stage0_b|base|SMASHX_CHAIN_STAGE=0|" \
  "g1024_stage1|base||--grid 1024" \
  "g1024_stage0|base|SMASHX_CHAIN_STAGE=0|--grid 1024" \
  "tile_stage1|base||--of 8 --as-rank 0" \
  "tile_stage0|base|SMASHX_CHAIN_STAGE=0|--of 8 --as-rank 0"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; rm -rf /tmp/st13
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st13 -- python3 bench.py --profile --steps 2 --warmup 1 > gpurun_out/s13_run.log 2>&1; echo "stats rc=$?"
f=$(find /tmp/st13 -name "*kernel_stats.csv" | head -1); head -1 "$f" > gpurun_out/s13_stats_2048.csv; grep -E "sx_k_" "$f" >> gpurun_out/s13_stats_2048.csv; grep -E "transpose|route" gpurun_out/s13_stats_2048.csv | cut -c1-160
