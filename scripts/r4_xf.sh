#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "operand_path" > $D/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $D/pytest.log; tail -30 $D/pytest.log | cut -c1-300
