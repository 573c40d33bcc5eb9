#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
timeout -k 10 900 python -m pytest tests/test_blocks_gpu.py tests/test_step_gpu.py -x -q -m gpu > $D/pytest_blocks.log 2>&1
echo "pytest blocks+step rc=$?"; tail -3 $D/pytest_blocks.log
for o in "conv_bnin=1" "conv_bnin=0"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt $o > $D/bench_$o.json 2> $D/bench_$o.err
  echo "== $o: $(python -c "import json;d=json.load(open('$D/bench_$o.json'));print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['conv_split'], d['config']['loss'])")"
done
