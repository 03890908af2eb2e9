#!/usr/bin/env bash
# Run on the GPU box: tools/profile_round.sh <tag> [bench args]
# Produces compact, committable summaries under gpurun_out/<tag>_* :
#   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats rows of the smashx kernels (same command as bench)
#   <tag>_kernel_trace_sx.csv per-dispatch rows of the smashx kernels (durations, VGPRs, grid)
#   <tag>_pmc.txt            FETCH_SIZE / WRITE_SIZE (+ SQ counters) per kernel, separate --pmc passes
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
raw=/tmp/prof_$tag
rm -rf $raw; mkdir -p $raw
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/stats -- python3 bench.py "$@" --no-cpu-baseline > $out/${tag}_stats_run.log 2>&1
f=$(find $raw/stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -1 "$f" > $out/${tag}_kernel_stats.csv; grep -E "sx_k_|k_gather|k_scatter|k_denorm|k_norm" "$f" >> $out/${tag}_kernel_stats.csv; fi
f=$(find $raw/stats -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then head -1 "$f" > $out/${tag}_kernel_trace_sx.csv; grep -E "sx_k_" "$f" >> $out/${tag}_kernel_trace_sx.csv; fi
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $raw/pmc$i -- python3 bench.py "$@" --no-cpu-baseline > $raw/pmc$i.log 2>&1 || echo "pmc group $i failed" >> $out/${tag}_errors.log
done
grep -h "^{" $out/${tag}_stats_run.log | tail -1 > $out/${tag}_bench_under_rocprof.json
cs=$(python3 -c "import json,sys; d=json.load(open('$out/${tag}_bench_under_rocprof.json')); print(d['config']['active_cells']*d['config']['nt'])" 2>/dev/null || echo 0)
python3 tools/pmc_summary.py $raw --json $out/${tag}_pmc_traffic.json --cellsteps $cs --command "rocprofv3 --kernel-trace --pmc <group> (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*) -- python3 bench.py $* --no-cpu-baseline (tools/profile_round.sh)" > $out/${tag}_pmc.txt 2>&1
