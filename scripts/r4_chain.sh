#!/bin/bash
D=gpurun_out/$1; mkdir -p $D
CAP=$PWD/very-large-scale-face-recognition_amd/libvlsfr_cap.so
for lib in base cap; do
  [ $lib = cap ] && export VLSFR_LIB=$CAP || unset VLSFR_LIB
  echo "== $lib" | tee -a $D/chain.txt
  python scripts/two_chain_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $D/chain.txt
  C=128 H=28 python scripts/two_chain_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $D/chain.txt
done
