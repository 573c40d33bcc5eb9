#!/bin/bash
# kernel durations (rocprofv3 kernel trace, not host-issue-bound timing loops) of scripts/bn_floor.py per kernel and grid size,
# under several OPTS settings: scripts/ktrace_bn.sh "<opts1>;<opts2>;..."
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra SETS <<< "$1"; shift
i=0
for o in "${SETS[@]}"; do
  i=$((i+1))
  export OPTS="$o"
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kb_$i -o k -- python3 $R/scripts/bn_floor.py "$@" > /tmp/kb_$i.log 2>&1 || { tail -n 5 /tmp/kb_$i.log; exit 1; }
  echo "== OPTS=[$o]"
  python3 - /tmp/kb_$i <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/k_kernel_trace.csv", recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if not ("bn_" in n or "copy" in n.lower()): continue
    g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
    agg.setdefault((n[:60], g), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(agg.items()):
    v.sort()
    if len(v) >= 10: print("%-62s grid %8d  n %4d  median %7.1f us  min %7.1f" % (n, g, len(v), v[len(v) // 2], v[0]))
PY
done
