#!/bin/bash
# the shapes of the packed region windows (C4GPU_WIN16: 1 = one wave per pair of chains, 5 ...: cooperating waves) on the
# north-star batch: agreement test, then one-lane and two-lane bench lines per shape
mkdir -p gpurun_out/win16
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -q -m gpu -x -k "region_windows or staged or north_star" > gpurun_out/win16/pytest.log 2>&1
tail -3 gpurun_out/win16/pytest.log
for w in ${WIN16_SHAPES:-1 5 6 7 8}; do
  for lanes in 1 2; do
    C4GPU_WIN16=$w C4GPU_LANES=$lanes timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs > gpurun_out/win16/bench_w${w}_l${lanes}.json 2> gpurun_out/win16/bench_w${w}_l${lanes}.err
    python3 - <<PY
import json
try:
    d = json.loads(open("gpurun_out/win16/bench_w${w}_l${lanes}.json").read().strip().splitlines()[-1])
    print("win16", "$w", "lanes", "$lanes", round(d["ms_per_step"], 1), "%.3e" % d["value"], {k: round(v, 1) for k, v in d.get("kernel_ms", {}).items()} if isinstance(d.get("kernel_ms"), dict) else "")
except Exception as e:
    print("win16 $w lanes $lanes failed", e)
PY
  done
done
