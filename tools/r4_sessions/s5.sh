#!/usr/bin/env bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf /tmp/st1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st1 -- python3 bench.py --profile --grid 1024 --steps 3 --warmup 1 > gpurun_out/s5_run.log 2>&1; echo "rc=$?"
f=$(find /tmp/st1 -name "*kernel_stats.csv" | head -1); head -1 "$f" > gpurun_out/s5_stats_1024.csv; grep -E "sx_k_" "$f" >> gpurun_out/s5_stats_1024.csv; cat gpurun_out/s5_stats_1024.csv | cut -c1-200
