#!/bin/bash
# 4-stage LDS ring of the default tiles for launches with at most one workgroup per CU (conv_deep_ring): batch 64
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
VLSFR_OPTIONS="conv_deep_ring=2" python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv or dgrad or bnred" > $D/pytest.log 2>&1
rc=$?; echo "pytest (conv_deep_ring=2) rc=$rc" | tee -a $D/pytest.log; tail -3 $D/pytest.log
[ $rc -ne 0 ] && exit $rc
for o in "conv_deep_ring=0" "conv_deep_ring=1" "conv_deep_ring=1 small_tile_wgs=256" "conv_deep_ring=0"; do
  echo "== $o" | tee -a $D/b64.txt
  ITERS=40 OPTS="$o" python scripts/conv_shapes.py 64 2>&1 | grep -v amdgpu.ids | tee -a $D/b64.txt
done
for o in "conv_deep_ring=0" "conv_deep_ring=1" "conv_deep_ring=0" "conv_deep_ring=1"; do
  python bench.py --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --opt $o > $D/bench64.json 2> $D/bench64.err
  echo "batch 64 $o: $(python -c "import json,sys; d=json.load(open('$D/bench64.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['achieved'], r['other']['conv_wgrad_kernel']['tflops'])")" | tee -a $D/bench64.txt
done
