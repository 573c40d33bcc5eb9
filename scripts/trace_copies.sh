#!/bin/bash
# Which HIP API calls make the __amd_rocclr_copyBuffer / fill launches of a step?  HIP API + kernel trace of a short bench;
# prints, for every hipMemcpyWithStream / hipMemcpyAsync call of the last traced step, the kernels launched just before it.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/copies
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --output-format csv -d /tmp/hc -o h -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serial --net ir18 --identities 65536 --batch 32 > $O/run.log 2>&1
T=$(find /tmp/hc -name "h_hip_api_trace.csv" | head -1)
python3 - "$T" > $O/sequence.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Function"] for r in rows]
idx = [i for i, n in enumerate(names) if n in ("hipMemcpyWithStream", "hipMemcpyAsync", "hipMemsetAsync")]
print(len(rows), "api calls;", len(idx), "copies/memsets")
# the last ~120 of them with the two preceding launches' positions
last = idx[-130:]
prev = None
for i in last:
    gap = i - prev if prev is not None else 0
    print(i, names[i], "launches since previous copy:", sum(1 for n in names[(prev or i) : i] if n == "hipLaunchKernel"))
    prev = i
PY
tail -n 140 $O/sequence.txt
