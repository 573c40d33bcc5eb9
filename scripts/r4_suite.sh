#!/bin/bash
# full GPU suite, then the default bench line (and a serial one)
D=gpurun_out/$1; mkdir -p $D
python -m pytest tests -x -q -m gpu > $D/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 $D/pytest.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $D/bench.json 2> $D/bench.err
echo "bench rc=$?"; cat $D/bench.json; grep "timed region done" $D/bench.err
