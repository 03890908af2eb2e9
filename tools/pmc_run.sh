#!/usr/bin/env bash
# Collect PMC counters for the bench kernels, one rocprofv3 pass per counter group (gfx950 slot limits:
# SQ 8, TCC 4 with FETCH_SIZE = 3 / WRITE_SIZE = 2; /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/pmc_run.sh <outdir> [bench args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/pmc$i" -- python3 bench.py "$@" --no-cpu-baseline > "$out/pmc$i.log" 2>&1 || echo "group $i failed" >> "$out/pmc_errors.log"
done
