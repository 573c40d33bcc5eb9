#!/bin/bash
# hw4 decomposition (no DMA / no epilogue / neither) and the 128-row tile on the 256-channel layers (conv_hp8=3)
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
O=$D/ab.txt
for o in "conv_dbg=0" "conv_hp8=3" "conv_dbg=32" "conv_dbg=64" "conv_dbg=96" "conv_dbg=0" "conv_hp8=3"; do
  echo "== $o" | tee -a $O
  ONLY="128_128_3_1_28 256_256_3_1_14" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $O
  echo "== $o NOSTATS" | tee -a $O
  NOSTATS=1 ONLY="128_128_3_1_28 256_256_3_1_14" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $O
done
