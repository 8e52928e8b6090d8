# round 3: shapes of the checkpoint pass (rows per lane x waves per SIMD the registers are held to): default 3 x 2; C = 2 x 3, D = 2 x 2, F = 1 x 4, G = 1 x 3
set -u
mkdir -p gpurun_out/r3p
for v in ${VARIANTS:-A D G C F A D G}; do
lib=exonerate_amd/libc4gpu.so; [ $v != A ] && lib=exonerate_amd/libc4gpu_$v.so
C4GPU_LIB=$PWD/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-revcomp --no-cpu-baseline > gpurun_out/r3p/bench_$v.json 2> gpurun_out/r3p/bench_$v.err; echo "bench $v rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3p/bench_$v.json').read().strip().splitlines()[-1])
print("$v ms_per_step %.1f" % d["ms_per_step"], {k: round(v/3,1) for k,v in d.get("kernel_ms",{}).items()}, "aligned_ok", d["config"].get("aligned_in_sample"))
PY
done
