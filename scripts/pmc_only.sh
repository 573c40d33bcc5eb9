R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${1:-r03}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial --counters-only > $O/pmc_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial --counters-only > $O/pmc_write.log 2>&1
echo "write pass done"
python3 $R/scripts/pmc_traffic.py $(find /tmp/pf -name 'f_counter_collection.csv' | head -1) $(find /tmp/pw -name 'w_counter_collection.csv' | head -1) $O/pmc_traffic.json ir100 256 10485760
