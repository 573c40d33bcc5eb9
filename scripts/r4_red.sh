#!/bin/bash
set -o pipefail
D=gpurun_out/$1; mkdir -p $D
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "bnred or halo_patch_four_phase" > $D/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $D/pytest.log; tail -3 $D/pytest.log
python -m pytest tests/test_blocks_gpu.py -x -q -m gpu > $D/pytest_blocks.log 2>&1
echo "pytest blocks rc=$?"; tail -2 $D/pytest_blocks.log
./scripts/probes/graph_memset_repro.bin > $D/graph_memset_repro.txt 2>&1; tail -4 $D/graph_memset_repro.txt
for o in "hw4_red=1" "hw4_red=0" "conv_hp8=0"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt $o > $D/bench_$o.json 2> $D/bench_$o.err
  echo "== $o: $(python -c "import json;d=json.load(open('$D/bench_$o.json'));print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('conv_split'))")"
done
