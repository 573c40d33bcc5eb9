#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4b
O=gpurun_out/r4b/decomp.txt
for o in "conv_hp8=1" "conv_hp8=1 conv_dbg=256" "conv_hp8=1 conv_dbg=32" "conv_hp8=1 conv_dbg=64" "conv_hp8=1 conv_dbg=96" "conv_hp8=1 conv_dbg=352" "conv_hp8=0 conv_dbg=32" "conv_hp8=0 conv_dbg=96"; do
  echo "== $o" | tee -a $O
  ONLY="128_128_3_1_28 256_256_3_1_14" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $O
done
echo "== trace 256ch" | tee -a $O
python scripts/hp8_trace.py 256 256 14 2>&1 | grep -v amdgpu.ids | tee -a $O
echo "== trace 128ch" | tee -a $O
python scripts/hp8_trace.py 256 128 28 2>&1 | grep -v amdgpu.ids | tee -a $O
