#!/bin/bash
# round 4, call c: the staged packed score pass (IO 1) against the per-step-load form and against the library before it
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4c; mkdir -p $OUT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs"
C4GPU_LIB=$ROOT/exonerate_amd/alt/libc4gpu_exp1.so C4GPU_LANES=1 timeout 600 $B > $OUT/exp1_lanes1.json 2> $OUT/exp1_lanes1.err
C4GPU_PK16_IO=0 C4GPU_LANES=1 timeout 600 $B > $OUT/io0_lanes1.json 2> $OUT/io0_lanes1.err
C4GPU_LANES=1 C4GPU_TRACE=1 timeout 600 $B > $OUT/io1_lanes1.json 2> $OUT/io1_lanes1.err
timeout 600 $B > $OUT/io1.json 2> $OUT/io1.err
timeout 1200 python -m pytest tests/test_gpu_kernel_variants.py -m gpu -x -q -k "packed or staged or window" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
grep -h 'workgroups per CU' $OUT/io1_lanes1.err | sort | uniq -c
for f in exp1_lanes1 io0_lanes1 io1_lanes1 io1; do python - <<P
import json
try:
    d=json.load(open("$OUT/$f.json")); print("$f", round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["kernel_ms"].items()})
except Exception as e: print("$f", "failed", e)
P
done
