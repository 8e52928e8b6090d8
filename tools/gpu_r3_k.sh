# round 3: the packed 16-bit score pass (two jobs per lane): parity of everything that goes through the windowed region pass, then the bench
set -u
mkdir -p gpurun_out/r3k
python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_configs.py -x -q -k "windowed or c4_north or shared" > gpurun_out/r3k/pytest.log 2>&1; echo "parity rc=$?"; tail -8 gpurun_out/r3k/pytest.log
for w in 1 0; do
C4GPU_PK16=$w python bench.py --steps 3 --warmup 1 --no-revcomp --no-cpu-baseline > gpurun_out/r3k/bench_pk$w.json 2> gpurun_out/r3k/bench_pk$w.err; echo "bench pk16=$w rc=$?"; tail -2 gpurun_out/r3k/bench_pk$w.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3k/bench_pk$w.json').read().strip().splitlines()[-1])
print("pk16=$w ms_per_step %.1f" % d["ms_per_step"], {k: round(v/3,1) for k,v in d["kernel_ms"].items()})
PY
done
