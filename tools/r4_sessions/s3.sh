#!/usr/bin/env bash
# round-4 GPU session 3: staging rows of the chained groups (A/B on one box), full GPU suite on the cleaned-up library
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s3_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/s3_pytest.log
python3 tools/ab_matrix.py --tag s3 --timeout 300 --steps 3 --warmup 1 -- \
  "stage1|base||" \
  "stage0|base|SMASHX_CHAIN_STAGE=0|" \
  "stage1_cs|base|SMASHX_CHAIN_STREAM=1|" \
  "stage1_b|base||" \
  "g1024_stage1|base||--grid 1024" \
  "g1024_stage0|base|SMASHX_CHAIN_STAGE=0|--grid 1024" \
  "tile_stage1|base||--of 8 --as-rank 0" \
  "tile_stage0|base|SMASHX_CHAIN_STAGE=0|--of 8 --as-rank 0" \
  "france_stage1|base||--mesh france:all" \
  "france_stage0|base|SMASHX_CHAIN_STAGE=0|--mesh france:all" \
  "d8_1024|base||--mesh d8 --grid 1024" \
  "fwdonly_1024|base||--grid 1024 --forward-only"
