#!/usr/bin/env bash
# round-4 GPU session 37: level tapes of the GR vertical kernels as one float4 per four steps (SX_TAPE_T4; variants/lib_t1.so = one row per step)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_exact.py -m gpu -x -q > gpurun_out/s37_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s37_pytest.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_matrix.py --tag s37 --timeout 300 --steps 3 --warmup 1 -- \
  "t4|base||" "t1|t1||" "t4_b|base||" "t1_b|t1||" \
  "g1024_t4|base||--grid 1024" "g1024_t1|t1||--grid 1024" \
  "grc_t4|base||--grid 1024 --structure gr-c" "grc_t1|t1||--grid 1024 --structure gr-c" \
  "tile_t4|base||--of 8 --as-rank 0" "tile_t1|t1||--of 8 --as-rank 0"
