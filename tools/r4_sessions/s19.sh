#!/usr/bin/env bash
# round-4 GPU session 19: first chained round (SMASHX_CHAIN_FROM) on the final schedule; the other structures at 1024^2 on the final build
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 tools/ab_matrix.py --tag s19 --timeout 300 --steps 3 --warmup 1 -- \
  "from1|base||" \
  "from2|base|SMASHX_CHAIN_FROM=2|" \
  "from3|base|SMASHX_CHAIN_FROM=3|" \
  "from1_b|base||" \
  "from2_b|base|SMASHX_CHAIN_FROM=2|" \
  "g1024_from1|base||--grid 1024" \
  "g1024_from2|base|SMASHX_CHAIN_FROM=2|--grid 1024" \
  "g1024_from2_stage1|base|SMASHX_CHAIN_FROM=2 SMASHX_CHAIN_STAGE=1|--grid 1024" \
  "tile_from1|base||--of 8 --as-rank 0" \
  "tile_from2|base|SMASHX_CHAIN_FROM=2|--of 8 --as-rank 0" \
  "fr_from1|base||--mesh france:all" \
  "fr_from2|base|SMASHX_CHAIN_FROM=2|--mesh france:all" \
  "gr_a|base||--grid 1024 --structure gr-a" \
  "gr_c|base||--grid 1024 --structure gr-c" \
  "gr_d|base||--grid 1024 --structure gr-d" \
  "vic_a|base||--grid 1024 --structure vic-a" \
  "d8_1024|base||--grid 1024 --mesh d8"
