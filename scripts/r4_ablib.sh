#!/bin/bash
# same-box A/B of two builds of the library: the in-tree libvlsfr.so ("new") against very-large-scale-face-recognition_amd/libvlsfr_old.so ("old")
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
OLD=$PWD/very-large-scale-face-recognition_amd/libvlsfr_old.so
SH="${SHAPES:-128_128_3_1_28 256_256_3_1_14}"
for rep in 1 2; do
  for lib in old new; do
    [ $lib = old ] && export VLSFR_LIB=$OLD || unset VLSFR_LIB
    echo "== $lib $rep" | tee -a $D/shapes.txt
    ONLY="$SH" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $D/shapes.txt
  done
done
for rep in 1 2; do
  for lib in old new; do
    [ $lib = old ] && export VLSFR_LIB=$OLD || unset VLSFR_LIB
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $D/bench_${lib}_$rep.json 2> $D/bench_${lib}_$rep.err
    echo "$lib $rep: $(python -c "import json,sys; d=json.load(open('$D/bench_${lib}_$rep.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['conv_split']['plain']['tflops'], r['conv_split']['with_bn_backward_reduction']['tflops'])")" | tee -a $D/bench.txt
  done
done
