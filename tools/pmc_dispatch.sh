#!/usr/bin/env bash
# usage (GPU box): tools/pmc_dispatch.sh <tag> "<counters>" <kernel-substring> [bench args]
# -> gpurun_out/<tag>_pmcd.txt : one line per dispatch of matching kernels with the requested counters
set -u
tag=$1; ctr=$2; pat=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
raw=/tmp/pmcd_$tag; rm -rf $raw
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $raw -- python3 bench.py "$@" --no-cpu-baseline > /tmp/pmcd_$tag.log 2>&1
f=$(find $raw -name "*counter_collection.csv" | head -1)
python3 - "$f" "$pat" > gpurun_out/${tag}_pmcd.txt <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
d=collections.OrderedDict()
for r in rows:
    k=(r['Dispatch_Id'], r['Kernel_Name'].split('(')[0].replace('void ','')[:30], r['Grid_Size'])
    d.setdefault(k,{})[r['Counter_Name']]=float(r['Counter_Value'])
for k,v in d.items():
    print(k[1].ljust(30), k[2].rjust(8), ' '.join('%s=%.4g'%(c,x) for c,x in sorted(v.items())))
PY
