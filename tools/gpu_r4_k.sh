#!/bin/bash
# round 4, call k: the whole GPU suite with its slowest tests listed, then smoke()
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4k; mkdir -p $OUT
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=40 > $OUT/pytest_gpu.log 2>&1
tail -60 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
