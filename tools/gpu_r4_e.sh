#!/bin/bash
# round 4, call e: the staged packed score pass with its profile reads in mid-step: timing, SQ counter breakdown (one launch lane), tests
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4e; mkdir -p $OUT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs"
C4GPU_LANES=1 timeout 600 $B > $OUT/io1_lanes1.json 2> $OUT/io1_lanes1.err
timeout 600 $B > $OUT/io1.json 2> $OUT/io1.err
for f in io1_lanes1 io1; do python - <<P
import json
try:
    d=json.load(open("$OUT/$f.json")); print("$f", round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["kernel_ms"].items()})
except Exception as e: print("$f", "failed", e)
P
done
C4GPU_LANES=1 bash tools/profile_sq_breakdown.sh r04_b > $OUT/sq.log 2>&1
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_kernel_variants.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
