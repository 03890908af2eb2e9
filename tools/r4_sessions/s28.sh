#!/usr/bin/env bash
# round-4 GPU session 28: macro-steps between two publications of a chained group's progress (SX_PK 4 / 8 / 16 / 32: variants/lib_pk*.so)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 tools/ab_matrix.py --tag s28 --timeout 300 --steps 3 --warmup 1 -- \
  "pk16|base||" \
  "pk8|pk8||" \
  "pk4|pk4||" \
  "pk32|pk32||" \
  "pk16_b|base||" \
  "pk8_b|pk8||" \
  "tile_pk16|base||--of 8 --as-rank 0" \
  "tile_pk8|pk8||--of 8 --as-rank 0" \
  "tile_pk4|pk4||--of 8 --as-rank 0" \
  "g1024_pk16|base||--grid 1024" \
  "g1024_pk8|pk8||--grid 1024" \
  "g1024_pk4|pk4||--grid 1024"
