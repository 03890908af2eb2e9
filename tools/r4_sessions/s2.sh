#!/usr/bin/env bash
# round-4 GPU session 2: chained rounds on their own stream / persistent, routing waves at raised priority; one kernel trace
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 420 python3 -m pytest tests -m gpu -x -q > gpurun_out/s2_pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/s2_pytest.log
SMASHX_CHAIN_STREAM=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/s2_pytest_cs.log 2>&1; echo "pytest(cs) rc=$?"
tail -3 gpurun_out/s2_pytest_cs.log
F6=24576; A4=26624
CAPS="SMASHX_VLDS_FWD=$F6 SMASHX_VLDS_ADJ=$A4"
python3 tools/ab_matrix.py --tag s2 --timeout 300 --steps 3 --warmup 1 -- \
  "base|base||" \
  "g256hp|g256hp||--group 256" \
  "g256h_p2_cs|g256h|$CAPS SMASHX_CHAIN_STREAM=1|--group 256 --pipe 1104" \
  "g256h_p4_cs|g256h|$CAPS SMASHX_CHAIN_STREAM=1|--group 256 --pipe 560" \
  "g256hp_p2_cs|g256hp|$CAPS SMASHX_CHAIN_STREAM=1|--group 256 --pipe 1104" \
  "g256hp_p4_cs|g256hp|$CAPS SMASHX_CHAIN_STREAM=1|--group 256 --pipe 560" \
  "g256hp_p8_cs|g256hp|$CAPS SMASHX_CHAIN_STREAM=1|--group 256 --pipe 280" \
  "g256hp_p4_cs_nocap|g256hp|SMASHX_CHAIN_STREAM=1|--group 256 --pipe 560" \
  "g256hp_p2_pers|g256hp|$CAPS SMASHX_PERSIST=1|--group 256 --pipe 1104" \
  "g256hp_p4_pers|g256hp|$CAPS SMASHX_PERSIST=1|--group 256 --pipe 560" \
  "g256hp_p4_pers64|g256hp|$CAPS SMASHX_PERSIST=1 SMASHX_PERSIST_WGS=64|--group 256 --pipe 560" \
  "prio_p2_cs|prio|SMASHX_CHAIN_STREAM=1|--pipe 1104" \
  "base_p2_cs|base|SMASHX_CHAIN_STREAM=1|--pipe 1104"
# one kernel trace of a pipelined configuration (timeline of the two streams)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tr1
env $CAPS SMASHX_CHAIN_STREAM=1 SMASHX_LIB=$PWD/variants/lib_g256hp.so timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -- python3 bench.py --profile --steps 2 --warmup 1 --group 256 --pipe 560 > gpurun_out/s2_trace_run.log 2>&1; echo "trace rc=$?"
f=$(find /tmp/tr1 -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then head -1 "$f" > gpurun_out/s2_trace_g256hp_p4_cs.csv; grep -E "sx_k_" "$f" | tail -400 >> gpurun_out/s2_trace_g256hp_p4_cs.csv; fi
