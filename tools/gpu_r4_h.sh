#!/bin/bash
# round 4, call h: the intron length counter as a sign test (score pass), tests that failed on their own expectations, C5's heuristic leg traced
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4h; mkdir -p $OUT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs"
C4GPU_LANES=1 timeout 600 $B > $OUT/lanes1.json 2> $OUT/lanes1.err
timeout 600 $B > $OUT/lanes2.json 2> $OUT/lanes2.err
for f in lanes1 lanes2; do python - <<P
import json
try:
    d=json.load(open("$OUT/$f.json")); print("$f", round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["kernel_ms"].items()})
except Exception as e: print("$f", "failed", e)
P
done
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -m gpu -x -q -k "staged or packed" > $OUT/pytest_kv.log 2>&1
tail -3 $OUT/pytest_kv.log
timeout 900 python -m pytest tests/test_integration_gpu.py -m gpu -x -q -k "ordinary_exit" > $OUT/pytest_exit.log 2>&1
tail -3 $OUT/pytest_exit.log
timeout 600 python tools/trace_c5_heuristic.py 256 $OUT/c5t > $OUT/c5_trace.txt 2>&1
rm -f $OUT/c5t/*.fa
cat $OUT/c5_trace.txt | head -80
