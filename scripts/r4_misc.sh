#!/bin/bash
D=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $D
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "graph" > $D/pytest_graph.log 2>&1
echo "pytest graph rc=$?"; tail -3 $D/pytest_graph.log
bash scripts/pmc_full_once.sh > $D/pmc_full_once.txt 2>&1
cp gpurun_out/pmc_full/status.txt $D/pmc_full_status.txt; cp gpurun_out/pmc_full/pmc_fetch_full.log $D/pmc_fetch_full.log
cat $D/pmc_full_status.txt; grep -c "^\[maps\]" $D/pmc_fetch_full.log; grep -E "SIGSEGV|Aborted|rocprofiler|libhsa|libvlsfr" $D/pmc_fetch_full.log | head -20
