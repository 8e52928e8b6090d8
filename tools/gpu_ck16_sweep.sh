#!/bin/bash
# the shapes of the packed checkpoint pass (C4GPU_CK16: 1 = one wave per pair of jobs, 5 ...: cooperating waves) and of the
# packed region windows (C4GPU_WIN16) on the north-star batch: agreement tests, then one-lane and two-lane bench lines per shape
mkdir -p gpurun_out/ck16
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout 1200 python -m pytest tests/test_gpu_kernel_variants.py -q -m gpu -x -k "checkpoint_pass or region_windows" > gpurun_out/ck16/pytest.log 2>&1
  tail -3 gpurun_out/ck16/pytest.log
fi
for spec in ${CK16_SHAPES:-1:1 5:1 6:1 7:1 5:5 5:8 6:8}; do
  set -- ${spec/:/ }
  for lanes in 1 2 2; do
    C4GPU_CK16=$1 C4GPU_WIN16=$2 C4GPU_LANES=$lanes timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs > gpurun_out/ck16/bench.json 2> gpurun_out/ck16/bench_c$1_w$2_l${lanes}.err
    python3 - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ck16/bench.json").read().strip().splitlines()[-1])
    print("ck16", "$1", "win16", "$2", "lanes", "$lanes", round(d["ms_per_step"], 1), "%.3e" % d["value"], {k: round(v, 1) for k, v in d.get("kernel_ms", {}).items()} if isinstance(d.get("kernel_ms"), dict) else "")
except Exception as e:
    print("ck16 $1 win16 $2 lanes $lanes failed", e)
PY
  done
done
