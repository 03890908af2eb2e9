#!/usr/bin/env bash
# round-4 GPU session 25: chunk buffers released to the V stream right after the staging copy (SMASHX_EARLY_RELEASE): parity, A/B
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py tests/test_gpu_rccl.py -m gpu -x -q > gpurun_out/s25_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s25_pytest.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_matrix.py --tag s25 --timeout 300 --steps 3 --warmup 1 -- \
  "early1|base||" \
  "early0|base|SMASHX_EARLY_RELEASE=0|" \
  "early1_b|base||" \
  "early0_b|base|SMASHX_EARLY_RELEASE=0|" \
  "early1_c|base||" \
  "early0_c|base|SMASHX_EARLY_RELEASE=0|" \
  "tile_early1|base||--of 8 --as-rank 0" \
  "tile_early0|base|SMASHX_EARLY_RELEASE=0|--of 8 --as-rank 0" \
  "g1024_stage1_early1|base|SMASHX_CHAIN_STAGE=1|--grid 1024 --chunk 2192" \
  "g1024_stage1_early0|base|SMASHX_CHAIN_STAGE=1 SMASHX_EARLY_RELEASE=0|--grid 1024 --chunk 2192"
