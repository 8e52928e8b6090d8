#!/bin/bash
# round 4, call p: plane-major column stage (no bank conflicts): step time, agreement tests
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4p; mkdir -p $OUT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs"
C4GPU_LANES=1 timeout 600 $B > $OUT/lanes1.json 2> $OUT/lanes1.err
timeout 600 $B > $OUT/lanes2.json 2> $OUT/lanes2.err
for f in lanes1 lanes2; do python - <<P
import json
try:
    d=json.load(open("$OUT/$f.json")); print("$f", round(d["ms_per_step"],1), {k: round(v/d["steps"],1) for k,v in d["kernel_ms"].items()})
except Exception as e: print("$f", "failed", e)
P
done
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -m gpu -x -q -k "staged or packed" > $OUT/pytest_kv.log 2>&1
tail -2 $OUT/pytest_kv.log
