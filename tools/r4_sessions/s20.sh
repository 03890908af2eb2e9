#!/usr/bin/env bash
# round-4 GPU session 20: the un-chained reverse routing kernel requests its inputs two macro-steps ahead (SX_PD_A = 2; variants/lib_pd1.so
# = the same sources with -DSX_PD_A=1): GPU suite, A/B on one box, kernel stats
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/s20_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/s20_pytest.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ab_matrix.py --tag s20 --timeout 300 --steps 3 --warmup 1 -- \
  "pd2|base||" \
  "pd1|pd1||" \
  "pd2_b|base||" \
  "pd1_b|pd1||" \
  "g1024_pd2|base||--grid 1024" \
  "g1024_pd1|pd1||--grid 1024" \
  "tile_pd2|base||--of 8 --as-rank 0" \
  "tile_pd1|pd1||--of 8 --as-rank 0" \
  "fr_pd2|base||--mesh france:all" \
  "fr_pd1|pd1||--mesh france:all"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; rm -rf /tmp/st20
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st20 -- python3 bench.py --profile --steps 2 --warmup 1 > gpurun_out/s20_run.log 2>&1; echo "stats rc=$?"
f=$(find /tmp/st20 -name "*kernel_stats.csv" | head -1); head -1 "$f" > gpurun_out/s20_stats_2048.csv; grep -E "sx_k_" "$f" >> gpurun_out/s20_stats_2048.csv; grep route gpurun_out/s20_stats_2048.csv | cut -c1-150
