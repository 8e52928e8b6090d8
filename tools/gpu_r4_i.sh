#!/bin/bash
# round 4, call i: C5's heuristic leg with every strand scanned on the device and one SDP flush; the drop-in's tests; dump spacing
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4i; mkdir -p $OUT
timeout 600 python tools/trace_c5_heuristic.py 256 $OUT/c5t > $OUT/c5_trace.txt 2>&1
rm -f $OUT/c5t/*.fa
grep -v '^  \*\*' $OUT/c5_trace.txt | head -40
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs"
for ks in 13 12 11; do
  C4GPU_SEED_KSHIFT=$ks timeout 600 $B > $OUT/ks$ks.json 2> $OUT/ks$ks.err
  python - <<P
import json
try:
    d=json.load(open("$OUT/ks$ks.json")); print("kshift $ks", round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["kernel_ms"].items()})
except Exception as e: print("kshift $ks", "failed", e)
P
done
timeout 1500 python -m pytest tests/test_integration_gpu.py -m gpu -x -q > $OUT/pytest_integration.log 2>&1
tail -3 $OUT/pytest_integration.log
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -m gpu -x -q -k "staged or packed" > $OUT/pytest_kv.log 2>&1
tail -3 $OUT/pytest_kv.log
