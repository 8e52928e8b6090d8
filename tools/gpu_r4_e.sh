# round 4: profiles of the library with the packed windows and the rooted checkpoint pass: one launch lane first (per-kernel
# times and counters add up to the step), then the default two lanes; dump spacing experiment; the configs block
set -u
mkdir -p gpurun_out/r4e
C4GPU_LANES=1 BENCH_EXTRA="--no-configs --no-revcomp --no-cpu-baseline" bash tools/profile_round.sh r04_a_lanes1 > gpurun_out/r4e/prof_lanes1.log 2>&1
bash tools/profile_round.sh r04_a > gpurun_out/r4e/prof.log 2>&1
tail -1 gpurun_out/prof_r04_a/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['value'], d.get('revcomp',{}).get('value'))
print(json.dumps(d.get('configs'), indent=1)[:3000])
print(d.get('ranks'))
"
for k in 12 11; do
  echo "== C4GPU_SEED_KSHIFT=$k"
  C4GPU_SEED_KSHIFT=$k timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs > gpurun_out/r4e/bench_k$k.json 2> gpurun_out/r4e/bench_k$k.err
  tail -1 gpurun_out/r4e/bench_k$k.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['value'], d['kernel_ms'])"
done
