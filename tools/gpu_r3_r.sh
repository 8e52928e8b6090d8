# round 3: default bench (both strands) with the hop budget following the longest way back
set -u
mkdir -p gpurun_out/r3r2
for rep in 1 2; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3r2/bench_$rep.json 2> gpurun_out/r3r2/bench_$rep.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3r2/bench_$rep.json').read().strip().splitlines()[-1])
print("default ms_per_step %.1f" % d["ms_per_step"], "revcomp ms %.1f value %.3e" % (d["revcomp"]["ms_per_step"], d["revcomp"]["value"]))
PY
done
