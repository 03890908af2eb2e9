#!/usr/bin/env bash
# round-4 GPU session 14: groups packed and exchange series numbered in the order of the inlets that read them (sx_plan.cpp):
# whole GPU suite on the new schedule, then A/B on one box against the previous packing (variants/lib_oldpack.so = the same sources
# with the previous sx_plan.cpp), staging rows on and off; kernel stats at 2048^2
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/s14_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s14_pytest.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_matrix.py --tag s14 --timeout 300 --steps 3 --warmup 1 -- \
  "new_stage1|base||" \
  "new_stage0|base|SMASHX_CHAIN_STAGE=0|" \
  "old_stage0|oldpack|SMASHX_CHAIN_STAGE=0|" \
  "old_stage1|oldpack||" \
  "new_stage1_b|base||" \
  "new_stage0_b|base|SMASHX_CHAIN_STAGE=0|" \
  "old_stage0_b|oldpack|SMASHX_CHAIN_STAGE=0|" \
  "g1024_new_stage1|base||--grid 1024" \
  "g1024_new_stage0|base|SMASHX_CHAIN_STAGE=0|--grid 1024" \
  "g1024_old_stage0|oldpack|SMASHX_CHAIN_STAGE=0|--grid 1024" \
  "tile_new_stage1|base||--of 8 --as-rank 0" \
  "tile_new_stage0|base|SMASHX_CHAIN_STAGE=0|--of 8 --as-rank 0" \
  "tile_old_stage0|oldpack|SMASHX_CHAIN_STAGE=0|--of 8 --as-rank 0"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; rm -rf /tmp/st14
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st14 -- python3 bench.py --profile --steps 2 --warmup 1 > gpurun_out/s14_run.log 2>&1; echo "stats rc=$?"
f=$(find /tmp/st14 -name "*kernel_stats.csv" | head -1); head -1 "$f" > gpurun_out/s14_stats_2048.csv; grep -E "sx_k_" "$f" >> gpurun_out/s14_stats_2048.csv; cut -c1-170 gpurun_out/s14_stats_2048.csv
