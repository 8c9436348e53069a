#!/bin/bash
# kernel-only times of one conv_bench shape under rocprofv3 for a list of library variants (on the GPU box)
# usage: bash tools/kernel_times.sh "<shape filter>" <which> <kernel regex> lib1 lib2 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; shape=$1; which=$2; rx=$3; shift 3
for lib in "$@"; do
  rm -rf /tmp/kt
  FEDFR_HIP_LIB_NAME=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/tools/conv_bench.py 30 "$shape" $which > /tmp/kt.log 2>&1
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; python3 -c "
import csv,re,sys
for r in csv.DictReader(open('$f')):
    if re.search(r'$rx', r['Name']): print('   %-70s calls %5s avg %7.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
"
done
