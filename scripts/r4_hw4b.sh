#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "halo_patch_four_phase" > $D/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $D/pytest.log; tail -3 $D/pytest.log
O=$D/ab.txt
for o in "conv_hp8=1 conv_hw4=1" "conv_hp8=0"; do
  echo "== $o" | tee -a $O
  ONLY="128_128_3_1_28 256_256_3_1_14" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $O
  echo "== $o NOSTATS" | tee -a $O
  NOSTATS=1 ONLY="128_128_3_1_28 256_256_3_1_14" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $O
done
echo "== trace 256ch" | tee -a $O
python scripts/hw4_trace.py 256 256 14 2>&1 | grep -v amdgpu.ids | tee -a $O
echo "== trace 128ch" | tee -a $O
python scripts/hw4_trace.py 256 128 28 2>&1 | grep -v amdgpu.ids | tee -a $O
