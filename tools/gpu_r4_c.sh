# round 4: the packed 16-bit checkpoint pass measured again (the first session's outputs were lost with its container):
# agreement tests, bench per shape (default steps), one-lane trace of the step
set -u
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -x -q -m gpu -k "packed_16_bit or device_route" 2>&1 | tail -8
for ck in 0 1 2 3 4 5; do
  echo "== C4GPU_CK16=$ck"
  C4GPU_CK16=$ck timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4c/bench_ck$ck.json 2> gpurun_out/r4c/bench_ck$ck.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4c/bench_ck$ck.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("revcomp"), {k: v for k, v in d.items() if "kernel" in k})
PY
done
echo "== one lane, trace"
C4GPU_LANES=1 C4GPU_TRACE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-revcomp > gpurun_out/r4c/bench_l1.json 2> gpurun_out/r4c/bench_l1.err
tail -60 gpurun_out/r4c/bench_l1.err
cat gpurun_out/r4c/bench_l1.json | tail -1
