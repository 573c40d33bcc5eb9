#!/bin/bash
D=gpurun_out/$1; mkdir -p $D
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "operand_path" > $D/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $D/pytest.log
for i in 1 2 3; do timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "operand_path" 2>&1 | tail -1; done
OPTS="conv_bnin=1" python scripts/bnin_micro.py 256 2>&1 | grep -v amdgpu.ids | tee $D/bnin_micro.txt
