#!/bin/bash
# A/B: conv_igemm_hw4_kernel capped at 448 registers (64 of a SIMD's 512 left for waves of other kernels) against the 480-register build
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
CAP=$PWD/very-large-scale-face-recognition_amd/libvlsfr_cap.so
for rep in 1 2; do
  for lib in base cap; do
    [ $lib = cap ] && export VLSFR_LIB=$CAP || unset VLSFR_LIB
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA > $D/bench_${lib}_$rep.json 2> $D/bench_${lib}_$rep.err
    echo "$lib $rep: $(python -c "import json,sys; d=json.load(open('$D/bench_${lib}_$rep.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'])")"
  done
done
for lib in base cap; do
  [ $lib = cap ] && export VLSFR_LIB=$CAP || unset VLSFR_LIB
  echo "== $lib" | tee -a $D/shapes.txt
  ONLY="128_128_3_1_28 256_256_3_1_14" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $D/shapes.txt
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --phases 2>&1 | grep -i "phase\|pass\|backward\|head" | head -20 | tee -a $D/phases_$lib.txt
done
