#!/bin/bash
# the config table of DESIGN.md section 8c: one bench line per configuration, same box, back to back
D=gpurun_out/$1; mkdir -p $D; O=$D/config_table.jsonl; : > $O
run() { echo "== $*" >&2; python bench.py --no-cpu-baseline "$@" 2>$D/last.err | tee -a $O | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('   ', d['value'], 'faces/s', d['ms_per_step'], 'ms', r['kernel'], r['frac'], {k:(v.get('tflops'),v.get('mfma_frac')) for k,v in r['other'].items()})"; }
run --steps 10 --warmup 3
run --steps 10 --warmup 3 --head-dtype fp8
run --steps 10 --warmup 3 --loss AM
run --steps 10 --warmup 3 --loss SV
run --steps 10 --warmup 3 --batch 64
run --steps 10 --warmup 3 --batch 128
run --steps 10 --warmup 3 --net ir50 --identities 1048576
run --steps 10 --warmup 3 --net ir50 --identities 1048576 --batch 64
run --steps 10 --warmup 3 --net mobile --head-dtype fp8
run --steps 10 --warmup 3 --net mobile
run --steps 20 --warmup 5 --net mobile --feat 128 --identities 1000 --batch 32
run --steps 10 --warmup 3 --net r50 --identities 1048576 --batch 128
run --steps 10 --warmup 3 --force-dist
run --steps 10 --warmup 3 --rehearse-world 8
run --steps 10 --warmup 3 --rehearse-world 8 --identities 104857600 --batch 64
run --steps 10 --warmup 3 --rehearse-world 8 --identities 104857600 --batch 64 --head-dtype fp8
