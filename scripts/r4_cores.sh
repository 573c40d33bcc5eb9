#!/bin/bash
D=gpurun_out/$1; mkdir -p $D
CAP=$PWD/very-large-scale-face-recognition_amd/libvlsfr_cap.so
for lib in base cap; do
  [ $lib = cap ] && export VLSFR_LIB=$CAP || unset VLSFR_LIB
  for mode in copy bn bnp; do
    echo "== $lib $mode C=256" | tee -a $D/probe.txt
    MODE=$mode python scripts/coresident_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $D/probe.txt
  done
  echo "== $lib bn C=128 H=28" | tee -a $D/probe.txt
  MODE=bn C=128 H=28 python scripts/coresident_probe.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $D/probe.txt
done
