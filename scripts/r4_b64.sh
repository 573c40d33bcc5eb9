#!/bin/bash
# batch 64: per-shape table under the tile / pipeline-depth variants that exist
D=gpurun_out/$1; mkdir -p $D
for o in "conv_dbg=0" "small_tile_wgs=256" "conv_glds=6" "conv_glds=4" "conv_glds=14" "conv_glds=12" "hp8_fill=40" "conv_halo=2"; do
  echo "== $o" | tee -a $D/b64.txt
  ITERS=40 ONLY="128_128_3_1_28 256_256_3_1_14 512_512_3_1_7 64_64_3_1_56" OPTS="$o" python scripts/conv_shapes.py 64 2>&1 | grep -v amdgpu.ids | tee -a $D/b64.txt
done
