#!/bin/bash
# round 4, call l: the parity failure of the full suite (derived_protein2genome_end) alone, after the kernel-variant tests, repeated;
# the checkpoint pass with its scalar gate
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4l; mkdir -p $OUT
for rep in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "derived_protein2genome" > $OUT/parity_alone_$rep.log 2>&1; tail -1 $OUT/parity_alone_$rep.log
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/parity_all.log 2>&1; tail -1 $OUT/parity_all.log
timeout 1500 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/kv_then_parity.log 2>&1; tail -3 $OUT/kv_then_parity.log
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
