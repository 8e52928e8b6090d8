#!/bin/bash
# dump interval of the windowed region pass (C4GPU_SEED_KSHIFT: columns between dumps = 1 << k) on the north-star batch
mkdir -p gpurun_out/kshift
for k in ${KSHIFTS:-13 12 11}; do
  for lanes in 1 2 2; do
    C4GPU_SEED_KSHIFT=$k C4GPU_LANES=$lanes timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs > gpurun_out/kshift/bench.json 2> gpurun_out/kshift/bench_k${k}_l${lanes}.err
    python3 - <<PY
import json
try:
    d = json.loads(open("gpurun_out/kshift/bench.json").read().strip().splitlines()[-1])
    print("kshift", "$k", "lanes", "$lanes", round(d["ms_per_step"], 1), "%.3e" % d["value"], {k: round(v, 1) for k, v in d.get("kernel_ms", {}).items()} if isinstance(d.get("kernel_ms"), dict) else "")
except Exception as e:
    print("kshift $k lanes $lanes failed", e)
PY
  done
done
