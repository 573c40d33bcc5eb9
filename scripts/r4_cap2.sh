#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
OLD=$PWD/very-large-scale-face-recognition_amd/libvlsfr_old.so
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > $D/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $D/pytest.log; tail -3 $D/pytest.log
[ $rc -ne 0 ] && exit $rc
for lib in old new; do
  [ $lib = old ] && export VLSFR_LIB=$OLD || unset VLSFR_LIB
  echo "== $lib" | tee -a $D/chain.txt
  python scripts/two_chain_probe.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $D/chain.txt
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --phases 2>&1 | grep -i "segment\|phase " | tee -a $D/phases_$lib.txt
done
ITERS=40 bash scripts/r4_ablib.sh $1
