#!/usr/bin/env bash
# round-4 GPU session 26: levels of the river tree per super-step in the chained rounds (SMASHX_SUBLEVELS) on the final schedule
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 tools/ab_matrix.py --tag s26 --timeout 300 --steps 3 --warmup 1 -- \
  "u1|base||" \
  "u2|base|SMASHX_SUBLEVELS=2|" \
  "u4|base|SMASHX_SUBLEVELS=4|" \
  "u1_b|base||" \
  "u2_b|base|SMASHX_SUBLEVELS=2|" \
  "tile_u1|base||--of 8 --as-rank 0" \
  "tile_u2|base|SMASHX_SUBLEVELS=2|--of 8 --as-rank 0" \
  "tile_u4|base|SMASHX_SUBLEVELS=4|--of 8 --as-rank 0" \
  "g1024_u1|base||--grid 1024" \
  "g1024_u2|base|SMASHX_SUBLEVELS=2|--grid 1024"
