#!/bin/bash
D=gpurun_out/$1; mkdir -p $D
for o in "conv_hw4=1" "bn_repl=32" "bn_repl=16" "wgrad_target_wgs=384" "wgrad_target_wgs=256" "wgrad_group=3" "conv_hw4=1"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt $o > $D/bench_$o.json 2> $D/bench_$o.err
  echo "== $o: $(python -c "import json;d=json.load(open('$D/bench_$o.json'));print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['other']['conv_wgrad_kernel']['tflops'])")"
done
