# round 3: shapes of the packed 16-bit score pass: C4GPU_PK16 = 1 (4 rows, 4 waves, 3 waves/SIMD), 2 (4, 4, 2), 3 (2 rows, 8 waves, 3), 4 (2, 8, 4), 0 (32-bit)
set -u
mkdir -p gpurun_out/r3l
for w in 1 0; do
C4GPU_PK16=$w python bench.py --steps 3 --warmup 1 --no-revcomp --no-cpu-baseline > gpurun_out/r3l/bench_pk$w.json 2> gpurun_out/r3l/bench_pk$w.err; echo "bench pk16=$w rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3l/bench_pk$w.json').read().strip().splitlines()[-1])
print("pk16=$w ms_per_step %.1f" % d["ms_per_step"], {k: round(v/3,1) for k,v in d["kernel_ms"].items()})
PY
done
