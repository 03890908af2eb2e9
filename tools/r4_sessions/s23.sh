#!/usr/bin/env bash
# round-4 GPU session 23: routing groups of 256 / 384 slots on the final schedule (two / one-and-a-half groups per CU desynchronise the barriers)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 tools/ab_matrix.py --tag s23 --timeout 300 --steps 3 --warmup 1 -- \
  "g512|base||" \
  "g256|base||--group 256" \
  "g384|base||--group 384" \
  "g512_b|base||" \
  "g256_b|base||--group 256" \
  "g1024_g512|base||--grid 1024" \
  "g1024_g256|base||--grid 1024 --group 256" \
  "tile_g512|base||--of 8 --as-rank 0" \
  "tile_g256|base||--of 8 --as-rank 0 --group 256"
