#!/bin/bash
# The north-star step on one and on two launch lanes under one or more settings of an environment switch (run on the GPU box
# through gpurun):  tools/r06_ab.sh <tag> <VAR> "<value> [<value> ...]" [steps]
set -u
TAG=$1; VAR=$2; VALS=$3; STEPS=${4:-4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for L in 1 2; do
  for V in $VALS; do
    env C4GPU_LANES=$L $VAR=$V python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-revcomp --no-configs > $OUT/bench_l${L}_${V}.json 2> $OUT/bench_l${L}_${V}.err
    python - <<PY
import json
d=json.loads(open("$OUT/bench_l${L}_${V}.json").read().strip().split("\n")[-1])
k=d.get("kernel_ms", {})
print("lanes $L $VAR=$V ms_per_step %.1f" % d.get("ms_per_step"), "per step:", {a: round(b / $STEPS, 1) for a, b in k.items()})
PY
  done
done
