#!/bin/bash
# kernel-trace stats of the single-stream bench (metric config); usage: ktrace_serial.sh <tag> [extra bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --serial "$@" > $O/serial.log 2>&1
cp $(find /tmp/ks -name "s_kernel_stats.csv" | head -1) $O/bench_serial_kernel_stats.csv; grep "^{" $O/serial.log > $O/bench_serial_profiled.json
echo "serial trace done"
