#!/bin/bash
D=gpurun_out/$1; mkdir -p $D
for o in "bn_block_kb=64" "bn_block_kb=32" "bn_block_kb=128" "bn_block_kb=256" "bn_block_kb=64" "bn_repl=4" "bn_xcd=1"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt $o > $D/bench.json 2> $D/bench.err
  echo "$o: $(python -c "import json,sys; d=json.load(open('$D/bench.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'])")" | tee -a $D/bench.txt
done
