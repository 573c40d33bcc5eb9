#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "halo_patch_four_phase or dgrad_bnred or conv" > $D/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $D/pytest.log; tail -3 $D/pytest.log
[ $rc -ne 0 ] && exit $rc
for o in "hw4_persist=0" "hw4_persist=1" "hw4_persist=0" "hw4_persist=1"; do
  echo "== $o" | tee -a $D/shapes.txt
  ITERS=40 ONLY="128_128_3_1_28 64_64_3_1_112 256_256_3_1_14" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $D/shapes.txt
done
for o in "hw4_persist=0" "hw4_persist=1" "hw4_persist=0" "hw4_persist=1"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt $o > $D/bench.json 2> $D/bench.err
  echo "$o: $(python -c "import json,sys; d=json.load(open('$D/bench.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['conv_split']['plain']['tflops'], r['conv_split']['with_bn_backward_reduction']['tflops'])")" | tee -a $D/bench.txt
done
