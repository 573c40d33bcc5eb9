#!/bin/bash
# Round-2 finding: `rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --serial` (WITHOUT --counters-only: per-launch HIP event
# brackets active after the timed region) died with a segmentation fault.  The brackets now take their events from a pool
# created up front (csrc/conv.hip prof_pool_reserve): this runs that very command once and records how it ends.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_full
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pfull -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial --full-under-pmc > $O/pmc_fetch_full.log 2>&1
echo "exit status of the rocprofv3 --pmc pass without --counters-only: $?" | tee $O/status.txt
grep -c "conv_igemm" $(find /tmp/pfull -name "f_counter_collection.csv" | head -1) >> $O/status.txt 2>&1
tail -n 5 $O/pmc_fetch_full.log | cut -c1-400
if [ "$1" = "repro" ]; then   # the same pass with round 2's per-launch event creation (vlsfr_set_option prof_pool = 0)
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pold -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial --full-under-pmc --opt prof_pool=0 > $O/pmc_fetch_old_events.log 2>&1
  echo "exit status with per-launch hipEventCreate (round 2 behaviour): $?" | tee -a $O/status.txt
  tail -n 3 $O/pmc_fetch_old_events.log | cut -c1-300
fi
