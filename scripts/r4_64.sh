#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "halo_patch_four_phase or one_wave_per_simd or bnred" > $D/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $D/pytest.log; tail -3 $D/pytest.log
O=$D/ab.txt
for o in "hw4_64=1" "hw4_64=0"; do
  echo "== $o" | tee -a $O
  ONLY="64_64_3_1_112 64_64_3_1_56 128_128_3_1_28 256_256_3_1_14" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $O
done
for o in "hw4_64=1" "hw4_64=0"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt $o > $D/bench_$o.json 2> $D/bench_$o.err
  echo "== $o: $(python -c "import json;d=json.load(open('$D/bench_$o.json'));print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['conv_split'])")"
done
