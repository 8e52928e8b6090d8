# round 4: packed region windows (c4_win16_kernel.h) + rooted packed checkpoint pass: agreement tests, then bench per shape
set -u
mkdir -p gpurun_out/r4d
timeout 1500 python -m pytest tests/test_gpu_kernel_variants.py -x -q -m gpu -k "packed_16_bit or windowed or window or device_route or lanes" 2>&1 | tail -25
echo "== bench default"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4d/bench_default.json 2> gpurun_out/r4d/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4d/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("revcomp", {}).get("value"), d.get("kernel_ms"))
PY
for w in 2 3 4; do
  echo "== C4GPU_WIN16=$w"
  C4GPU_WIN16=$w timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-revcomp > gpurun_out/r4d/bench_w$w.json 2> gpurun_out/r4d/bench_w$w.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4d/bench_w$w.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("kernel_ms"))
PY
done
for ck in 2 3 4; do
  echo "== C4GPU_CK16=$ck"
  C4GPU_CK16=$ck timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-revcomp > gpurun_out/r4d/bench_ck$ck.json 2> gpurun_out/r4d/bench_ck$ck.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4d/bench_ck$ck.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("kernel_ms"))
PY
done
echo "== one lane, trace"
C4GPU_LANES=1 C4GPU_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-revcomp > gpurun_out/r4d/bench_l1.json 2> gpurun_out/r4d/bench_l1.err
grep -E "kernel \+ results|seeded pass|fused:|windowed|done  " gpurun_out/r4d/bench_l1.err | tail -28
