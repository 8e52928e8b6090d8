#!/bin/bash
# round 4, call b: instruction-class microbenchmark, the default bench line, a quick parity subset on the rebuilt tree
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4b; mkdir -p $OUT
timeout 600 build/inst_class_microbench > $OUT/inst_class.json 2> $OUT/inst_class.err
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
C4GPU_LANES=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs > $OUT/bench_lanes1.json 2> $OUT/bench_lanes1.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
