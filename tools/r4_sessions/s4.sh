#!/usr/bin/env bash
# round-4 GPU session 4: staging rows with the coalesced gather; early start of the next vertical kernel on / off
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s4_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/s4_pytest.log
python3 tools/ab_matrix.py --tag s4 --timeout 300 --steps 3 --warmup 1 -- \
  "stage1|base||" \
  "stage0|base|SMASHX_CHAIN_STAGE=0|" \
  "stage1_early0|base|SMASHX_EARLY_V=0|" \
  "stage1_b|base||" \
  "stage0_b|base|SMASHX_CHAIN_STAGE=0|" \
  "g1024_stage1|base||--grid 1024" \
  "g1024_stage0|base|SMASHX_CHAIN_STAGE=0|--grid 1024" \
  "tile_stage1|base||--of 8 --as-rank 0" \
  "tile_stage0|base|SMASHX_CHAIN_STAGE=0|--of 8 --as-rank 0" \
  "tile_stage1_p2192|base||--of 8 --as-rank 0 --pipe 2192" \
  "france_stage1|base||--mesh france:all" \
  "france_stage0|base|SMASHX_CHAIN_STAGE=0|--mesh france:all"
