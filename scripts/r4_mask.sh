#!/bin/bash
D=gpurun_out/$1; mkdir -p $D
for r in 0 32 0 32 16 64; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cu-reserve $r > $D/bench.json 2> $D/bench_$r.err || { tail -5 $D/bench_$r.err; continue; }
  echo "cu-reserve $r: $(python -c "import json,sys; d=json.load(open('$D/bench.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'])")" | tee -a $D/bench.txt
done
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --phases --cu-reserve 32 2>&1 | grep -i "segment\|phase " | tee -a $D/phases32.txt
