#!/usr/bin/env bash
# round-4 GPU session 1: suite on the new host code + first co-residency matrix at 2048^2
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 420 python3 -m pytest tests -m gpu -x -q > gpurun_out/s1_pytest.log 2>&1; echo "pytest rc=$?" 
tail -3 gpurun_out/s1_pytest.log
F6=24576; A4=26624; A3=34816
python3 tools/ab_matrix.py --tag s1 --timeout 300 --steps 3 --warmup 1 -- \
  "base|base||" \
  "base_caps|base|SMASHX_VLDS_FWD=$F6 SMASHX_VLDS_ADJ=$A4|" \
  "base_p2|base||--pipe 1104" \
  "g256|g256||--group 256" \
  "g256_p2|g256|SMASHX_VLDS_FWD=$F6 SMASHX_VLDS_ADJ=$A3|--group 256 --pipe 1104" \
  "g256_p4|g256|SMASHX_VLDS_FWD=$F6 SMASHX_VLDS_ADJ=$A3|--group 256 --pipe 560" \
  "g256h|g256h||--group 256" \
  "g256h_p2|g256h|SMASHX_VLDS_FWD=$F6 SMASHX_VLDS_ADJ=$A4|--group 256 --pipe 1104" \
  "g256h_p4|g256h|SMASHX_VLDS_FWD=$F6 SMASHX_VLDS_ADJ=$A4|--group 256 --pipe 560" \
  "g256m2|g256m2||--group 256" \
  "g256m2_p2|g256m2|SMASHX_VLDS_FWD=$F6 SMASHX_VLDS_ADJ=$A4|--group 256 --pipe 1104" \
  "g256m2_p4|g256m2|SMASHX_VLDS_FWD=$F6 SMASHX_VLDS_ADJ=$A4|--group 256 --pipe 560" \
  "g256h_p2_nocap|g256h||--group 256 --pipe 1104"
