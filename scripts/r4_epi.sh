#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
python -m pytest tests/test_ops_gpu.py -x -q -m gpu > $D/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $D/pytest.log; tail -3 $D/pytest.log
O=$D/ab.txt
for o in "conv_dbg=0" "conv_dbg=0"; do
  echo "== $o" | tee -a $O
  ONLY="128_128_3_1_28 256_256_3_1_14 64_64_3_1_56 512_512_3_1_7 256_256_3_2_28" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $O
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $D/bench.json 2> $D/bench.err
cat $D/bench.json | cut -c1-1200
