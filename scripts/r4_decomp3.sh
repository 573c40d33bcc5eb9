#!/bin/bash
# hw4 epilogue: all stores masked off (512), stores redirected to 256 cache-resident rows (1024), no epilogue at all (64)
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
O=$D/ab.txt
for o in "conv_dbg=0" "conv_dbg=512" "conv_dbg=1024" "conv_dbg=64" "conv_dbg=0" "conv_dbg=512" "conv_dbg=1024"; do
  echo "== $o" | tee -a $O
  ONLY="128_128_3_1_28 256_256_3_1_14" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $O
done
