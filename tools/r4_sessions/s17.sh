#!/usr/bin/env bash
# round-4 GPU session 17: round-0 groups in spatial blocks of 64 (sx_plan.cpp, SMASHX_BLOCK_ORDER): parity subset, then A/B on one box
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py -m gpu -x -q > gpurun_out/s17_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s17_pytest.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_matrix.py --tag s17 --timeout 300 --steps 3 --warmup 1 -- \
  "blk1|base||" \
  "blk0|base|SMASHX_BLOCK_ORDER=0|" \
  "blk1_b|base||" \
  "blk0_b|base|SMASHX_BLOCK_ORDER=0|" \
  "g1024_blk1|base||--grid 1024" \
  "g1024_blk0|base|SMASHX_BLOCK_ORDER=0|--grid 1024" \
  "fwd1024_blk1|base||--grid 1024 --forward-only" \
  "fwd1024_blk0|base|SMASHX_BLOCK_ORDER=0|--grid 1024 --forward-only" \
  "tile_blk1|base||--of 8 --as-rank 0" \
  "tile_blk0|base|SMASHX_BLOCK_ORDER=0|--of 8 --as-rank 0" \
  "fr_blk1|base||--mesh france:all" \
  "fr_blk0|base|SMASHX_BLOCK_ORDER=0|--mesh france:all"
