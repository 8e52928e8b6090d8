#!/bin/bash
# round 4, call f: staged packed score pass with dead rows silenced; readfirstlane in every persistent kernel; printers on device alignments
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4f; mkdir -p $OUT
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
timeout 1500 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_printers.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest2.log 2>&1
tail -3 $OUT/pytest2.log
