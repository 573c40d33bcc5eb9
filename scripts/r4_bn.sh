#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
OLD=$PWD/very-large-scale-face-recognition_amd/libvlsfr_old.so
python -m pytest tests/test_ops_gpu.py -x -q -m gpu > $D/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $D/pytest.log; tail -3 $D/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
for lib in old new; do
  [ $lib = old ] && export VLSFR_LIB=$OLD || unset VLSFR_LIB
  echo "== $lib $rep" | tee -a $D/bn.txt
  python scripts/bn_shapes.py 256 2>&1 | grep -v amdgpu.ids | tee -a $D/bn.txt
done
done
for rep in 1 2; do
  for lib in old new; do
    [ $lib = old ] && export VLSFR_LIB=$OLD || unset VLSFR_LIB
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $D/bench_${lib}_$rep.json 2> $D/bench_${lib}_$rep.err
    echo "$lib $rep: $(python -c "import json,sys; d=json.load(open('$D/bench_${lib}_$rep.json')); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'])")" | tee -a $D/bench.txt
  done
done
