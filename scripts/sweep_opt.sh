#!/bin/bash
# usage: sweep_opt.sh name v1 v2 ...   (runs bench.py with --opt name=v for each value)
name=$1; shift
for v in "$@"; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --opt $name=$v 2> /dev/null | grep "^{" > /tmp/sweep.json
  python -c "import json; d=json.load(open('/tmp/sweep.json')); print('$name=$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['other'])"
done
