# round 4: first run of the packed 16-bit checkpoint pass: agreement test, existing reduced-space tests, bench per shape
set -u
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -x -q -m gpu -k "packed_16_bit_checkpoint or device_route" 2>&1 | tail -15
for ck in 0 1 2 3 4 5; do
  echo "== C4GPU_CK16=$ck"
  C4GPU_CK16=$ck timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-revcomp > gpurun_out/r4a/bench_ck$ck.json 2> gpurun_out/r4a/bench_ck$ck.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4a/bench_ck$ck.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], {k: v for k, v in d.items() if "kernel" in k or "ms" in k})
PY
done
