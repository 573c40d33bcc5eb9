#!/bin/bash
# kernel trace of the default (three-stream) bench + timeline statistics
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o c -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --counters-only > $O/conc.log 2>&1
f=$(find /tmp/kt -name "c_kernel_trace.csv" | head -1)
python3 $R/scripts/trace_overlap.py $f | tee $O/overlap.txt
python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --phases 2>&1 | grep -E "phase|segment" | tee $O/phases.txt
