#!/usr/bin/env bash
# usage (GPU box): tools/trace_rounds.sh <tag> [bench args] -> gpurun_out/<tag>_rounds.txt : per-dispatch ms of the smashx kernels
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
raw=/tmp/trace_$tag; rm -rf $raw
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $raw -- python3 bench.py "$@" --no-cpu-baseline > /tmp/trace_$tag.log 2>&1
f=$(find $raw -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_rounds.txt <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'sx_k_' in r['Kernel_Name']]
for r in rows:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    print(r['Kernel_Name'].split('(')[0].replace('void ','')[:28].ljust(28), r['Grid_Size_X'].rjust(8), r['Workgroup_Size_X'].rjust(4), '%9.3f ms'%d, 'vgpr',r['VGPR_Count'],'lds',r['LDS_Block_Size'])
PY
