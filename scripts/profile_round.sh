#!/bin/bash
# Round evidence for the metric's own configuration (bench.py defaults: ir100 + 10 485 760 identities, batch_size 256):
# kernel-trace stats of bench.py (single stream and the three-stream schedule) and the two PMC traffic passes
# (FETCH_SIZE, WRITE_SIZE in separate runs with nothing but --pmc: see MI355X_MICROARCH.md, HBM / rocprofv3 sections).
# Writes under gpurun_out/prof_$1/ (copy what should be judged into profiles/).   usage: profile_round.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --serial > $O/serial.log 2>&1
cp $(find /tmp/ks -name "s_kernel_stats.csv" | head -1) $O/bench_serial_kernel_stats.csv; grep "^{" $O/serial.log > $O/bench_serial_profiled.json
echo "serial trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o c -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/conc.log 2>&1
cp $(find /tmp/kc -name "c_kernel_stats.csv" | head -1) $O/bench_concurrent_kernel_stats.csv; grep "^{" $O/conc.log > $O/bench_concurrent_profiled.json
echo "concurrent trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial --counters-only > $O/pmc_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial --counters-only > $O/pmc_write.log 2>&1
echo "write pass done"
python3 $R/scripts/pmc_traffic.py $(find /tmp/pf -name 'f_counter_collection.csv' | head -1) $(find /tmp/pw -name 'w_counter_collection.csv' | head -1) $O/pmc_traffic.json ir100 256 10485760
# config C5's shape with the fp8 class matmul (MobileFaceNet + 10 485 760 identities, batch_size 256): kernel-trace stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -o s -- python3 $R/bench.py --net mobile --head-dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline --serial > $O/c5.log 2>&1
cp /tmp/k5/s_kernel_stats.csv $O/c5_mobile_fp8_serial_kernel_stats.csv 2>/dev/null || cp $(find /tmp/k5 -name "s_kernel_stats.csv" | head -1) $O/c5_mobile_fp8_serial_kernel_stats.csv; grep "^{" $O/c5.log > $O/c5_mobile_fp8_serial_profiled.json
echo "c5 trace done"
