#!/usr/bin/env bash
# round-4 GPU session 29: smaller groups in the chained rounds only (SMASHX_GROUP_LATE): shorter super-steps of the latency chain
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SMASHX_GROUP_LATE=256 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py -m gpu -x -q > gpurun_out/s29_pytest.log 2>&1; rc=$?; echo "pytest (late groups of 256) rc=$rc"
tail -3 gpurun_out/s29_pytest.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_matrix.py --tag s29 --timeout 300 --steps 3 --warmup 1 -- \
  "l512|base||" \
  "l256|base|SMASHX_GROUP_LATE=256|" \
  "l128|base|SMASHX_GROUP_LATE=128|" \
  "l384|base|SMASHX_GROUP_LATE=384|" \
  "l512_b|base||" \
  "l256_b|base|SMASHX_GROUP_LATE=256|" \
  "tile_l512|base||--of 8 --as-rank 0" \
  "tile_l256|base|SMASHX_GROUP_LATE=256|--of 8 --as-rank 0" \
  "tile_l128|base|SMASHX_GROUP_LATE=128|--of 8 --as-rank 0" \
  "g1024_l512|base||--grid 1024" \
  "g1024_l256|base|SMASHX_GROUP_LATE=256|--grid 1024" \
  "g1024_l128|base|SMASHX_GROUP_LATE=128|--grid 1024" \
  "fr_l512|base||--mesh france:all" \
  "fr_l256|base|SMASHX_GROUP_LATE=256|--mesh france:all"
