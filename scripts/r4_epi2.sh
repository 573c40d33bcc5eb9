#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
python -m pytest tests/test_ops_gpu.py -x -q -m gpu > $D/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $D/pytest.log; tail -3 $D/pytest.log
[ $rc -ne 0 ] && exit $rc
bash scripts/r4_ablib.sh $1
