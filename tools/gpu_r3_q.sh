# round 3: hop budget of the chained region windows on the reverse strands (12 = until now; 16 covers every path of a 100 kb window at 8 192 columns per dump)
set -u
mkdir -p gpurun_out/r3q
for h in 12 16; do
C4GPU_WINDOW_HOPS=$h python tools/trace_revcomp.py > gpurun_out/r3q/rev_h$h.out 2> gpurun_out/r3q/rev_h$h.err; echo "rev hops=$h rc=$?"; cat gpurun_out/r3q/rev_h$h.out
grep -E "windowed region pass|kernel \+ results|run mode" gpurun_out/r3q/rev_h$h.err | tail -12
done
for h in 12 16; do
C4GPU_WINDOW_HOPS=$h python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3q/bench_h$h.json 2> gpurun_out/r3q/bench_h$h.err; echo "bench hops=$h rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3q/bench_h$h.json').read().strip().splitlines()[-1])
print("hops $h ms_per_step %.1f" % d["ms_per_step"], "revcomp ms %.1f value %.3e" % (d["revcomp"]["ms_per_step"], d["revcomp"]["value"]))
PY
done
