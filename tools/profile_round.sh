#!/usr/bin/env bash
# Run on the GPU box: tools/profile_round.sh <tag> [bench args]
# Profiles ONE process running ONE workload: `python3 bench.py --profile [bench args]` (the headline case alone -- no secondary
# case, no tile_solo, no exact-libm child, no CPU baseline, no inclusive calls).  --gpus is refused: a launcher or child process
# under the profiler would put several pids' kernels under the same sx_k_* names.
# Produces compact, committable summaries under gpurun_out/<tag>_* :
#   <tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats rows of the smashx kernels
#   <tag>_kernel_trace_sx.csv per-dispatch rows of the smashx kernels (durations, VGPRs, grid)
#   <tag>_pmc.txt / <tag>_pmc_traffic.json   FETCH_SIZE / WRITE_SIZE (+ SQ counters) per kernel family, separate --pmc passes,
#                             divided by the cell-steps bench.py says the profiled process swept (profile_accounting)
set -u
tag=$1; shift
for x in "$@"; do case "$x" in --gpus|--gpus=*) echo "profile_round.sh: --gpus is not allowed under the profiler" >&2; exit 2;; esac; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
raw=/tmp/prof_$tag
rm -rf $raw; mkdir -p $raw $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/stats -- python3 bench.py --profile "$@" > $out/${tag}_stats_run.log 2>&1 || { echo "stats pass failed" >> $out/${tag}_errors.log; exit 1; }
npid=$(find $raw/stats -name "*kernel_stats.csv" | wc -l)
if [ "$npid" != "1" ]; then echo "expected one profiled process, found $npid" >> $out/${tag}_errors.log; exit 1; fi
f=$(find $raw/stats -name "*kernel_stats.csv")
head -1 "$f" > $out/${tag}_kernel_stats.csv; grep -E "sx_k_|k_gather|k_scatter|k_denorm|k_norm|k_halo|k_encode" "$f" >> $out/${tag}_kernel_stats.csv
f=$(find $raw/stats -name "*kernel_trace.csv")
head -1 "$f" > $out/${tag}_kernel_trace_sx.csv; grep -E "sx_k_" "$f" >> $out/${tag}_kernel_trace_sx.csv
grep -h "^{" $out/${tag}_stats_run.log | tail -1 > $out/${tag}_bench_under_rocprof.json
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  # a GPU step that timed out or was killed ends the script: no further GPU step in this call
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $raw/pmc$i -- python3 bench.py --profile "$@" > $raw/pmc$i.log 2>&1 || { echo "pmc group $i failed" >> $out/${tag}_errors.log; tail -5 $raw/pmc$i.log >> $out/${tag}_errors.log; exit 1; }
done
# optional pass (its counters may not exist on every ROCm build): the executed fp64 / transcendental mix
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 --output-format csv -d $raw/pmc5 -- python3 bench.py --profile "$@" > $raw/pmc5.log 2>&1 || { echo "optional mix pass failed" >> $out/${tag}_errors.log; tail -5 $raw/pmc5.log >> $out/${tag}_errors.log; rm -rf $raw/pmc5; }
python3 tools/pmc_summary.py $raw --bench $out/${tag}_bench_under_rocprof.json --json $out/${tag}_pmc_traffic.json \
  --command "rocprofv3 --kernel-trace --pmc <group> (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_* | GRBM_GUI_ACTIVE) -- python3 bench.py --profile $* (tools/profile_round.sh)" > $out/${tag}_pmc.txt 2>&1
