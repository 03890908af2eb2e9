#!/usr/bin/env bash
# round-4 GPU session 33: copy passes with 1 / 2 / 4 / 6 consecutive wave-blocks per workgroup (SX_STG_WAVES; variants/lib_sw*.so)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py -m gpu -x -q -k "staging or chunk or real_river" > gpurun_out/s33_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s33_pytest.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_matrix.py --tag s33 --timeout 300 --steps 3 --warmup 1 -- \
  "sw4|base||" "sw1|sw1||" "sw2|sw2||" "sw6|sw6||" "sw4_b|base||" "sw1_b|sw1||" \
  "tile_sw4|base||--of 8 --as-rank 0" "tile_sw1|sw1||--of 8 --as-rank 0"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; rm -rf /tmp/st33
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st33 -- python3 bench.py --profile --steps 2 --warmup 1 > gpurun_out/s33_run.log 2>&1; echo "stats rc=$?"
f=$(find /tmp/st33 -name "*kernel_stats.csv" | head -1); grep -E "transpose" "$f" | cut -c1-150
