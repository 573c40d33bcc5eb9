#!/bin/bash
# round 4: conv_igemm_hp8_kernel parity + A/B against the round-3 kernels on the two layer shapes it targets
set -o pipefail
mkdir -p gpurun_out/r4c
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "halo_patch_four_phase or one_round_tiles or benchmarked_extents" > gpurun_out/r4c/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r4c/pytest.log
tail -5 gpurun_out/r4c/pytest.log
for o in "conv_hp8=0" "conv_hp8=1 conv_hw4=0" "conv_hp8=1 conv_hw4=1"; do
  echo "== $o" | tee -a gpurun_out/r4c/ab.txt
  ONLY="128_128_3_1_28 256_256_3_1_14" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | tee -a gpurun_out/r4c/ab.txt
done
for o in "conv_hp8=0" "conv_hp8=1 conv_hw4=0" "conv_hp8=1 conv_hw4=1"; do
  echo "== $o" | tee -a gpurun_out/r4c/ab.txt
  ONLY="128_128_3_1_28 256_256_3_1_14" OPTS="$o" python scripts/conv_shapes.py 256 2>&1 | tee -a gpurun_out/r4c/ab.txt
done
