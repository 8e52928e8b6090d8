# round 3: region windows chained on the device (one launch): parity tests of the Viterbi path + bench
set -u
mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py -x -q --durations=8 > gpurun_out/r3c/pytest.log 2>&1; echo "parity rc=$?"; tail -14 gpurun_out/r3c/pytest.log
C4GPU_TRACE=1 python bench.py --steps 5 --warmup 1 --no-revcomp --no-cpu-baseline > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c/bench.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], d["kernel_ms"], d["roofline"]["launches"])
PY
grep "windowed region" gpurun_out/r3c/bench.err | tail -3
