#!/bin/bash
D=gpurun_out/$1; mkdir -p $D
for a in "--batch 64" "--batch 64 --graphs" "--batch 64" "--batch 64 --graphs" "--batch 64 --graphs --opt conv_deep_ring=1" "--batch 128" "--batch 128 --graphs" "" "--graphs"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline $a > $D/bench.json 2> $D/bench.err || { echo "$a FAILED"; tail -3 $D/bench.err; continue; }
  echo "[$a]: $(python -c "import json,sys; d=json.load(open('$D/bench.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'])")" | tee -a $D/bench.txt
done
