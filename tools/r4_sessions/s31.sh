#!/usr/bin/env bash
# round-4 GPU session 31: wavefronts whose lanes are still OR in a data gap take the short form of the vertical step
# (variants/lib_before.so = the previous commit): parity of both builds, A/B
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_exact.py tests/test_gpu_tangent.py -m gpu -x -q > gpurun_out/s31_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s31_pytest.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_matrix.py --tag s31 --timeout 300 --steps 3 --warmup 1 -- \
  "new|base||" "old|before||" "new_b|base||" "old_b|before||" \
  "g1024_new|base||--grid 1024" "g1024_old|before||--grid 1024" \
  "fwd_new|base||--grid 1024 --forward-only" "fwd_old|before||--grid 1024 --forward-only" \
  "tile_new|base||--of 8 --as-rank 0" "tile_old|before||--of 8 --as-rank 0"
