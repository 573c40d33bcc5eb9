#!/bin/bash
D=gpurun_out/$1; mkdir -p $D
for o in "head_dma_spread=0" "head_dma_spread=1" "head_dma_spread=0" "head_dma_spread=1"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt $o > $D/bench_$o.json 2> $D/bench_$o.err
  echo "== $o: $(python -c "import json;d=json.load(open('$D/bench_$o.json'));print(d['ms_per_step'], d['value'], d['roofline']['other']['head_sweep_kernel'])")"
done
VLSFR_OPTIONS=head_dma_spread=1 timeout -k 10 600 python -m pytest tests/test_head_gpu.py -x -q -m gpu > $D/pytest_head.log 2>&1
echo "pytest head (spread=1) rc=$?"; tail -2 $D/pytest_head.log
