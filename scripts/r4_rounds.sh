#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
VLSFR_OPTIONS="hw4_rounds=1" python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "halo_patch_four_phase or dgrad_bnred" > $D/pytest.log 2>&1
rc=$?; echo "pytest (hw4_rounds=1) rc=$rc" | tee -a $D/pytest.log; tail -2 $D/pytest.log
[ $rc -ne 0 ] && exit $rc
for o in "hw4_rounds=0" "hw4_rounds=1" "hw4_rounds=0" "hw4_rounds=1"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt $o > $D/bench.json 2> $D/bench.err
  echo "$o: $(python -c "import json,sys; d=json.load(open('$D/bench.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'])")" | tee -a $D/bench.txt
done
