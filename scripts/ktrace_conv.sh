#!/bin/bash
# kernel-trace timing of conv_micro under several OPTS settings: scripts/ktrace_conv.sh "<opts1>;<opts2>;..." <micro args...>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra SETS <<< "$1"; shift
i=0
for o in "${SETS[@]}"; do
  i=$((i+1))
  export OPTS="$o"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$i -o k -- python $R/scripts/conv_micro.py "$@" > /tmp/kt_$i.log 2>&1
  echo "OPTS=[$o] NOSTATS=${NOSTATS:-} :: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('/tmp/kt_$i/k_kernel_stats.csv')):
    if 'conv_' in r['Name']: print(r['Name'].split('::')[-2][:40] if False else r['Name'][28:70], r['Calls'], 'avg_us=%.1f min_us=%.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3), end='  ')
")"
done
