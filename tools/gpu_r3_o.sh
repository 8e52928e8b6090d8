# round 3: dump rows of the packed score pass as dwordx4 stores (libc4gpu.so) against single-dword stores (libc4gpu_B.so, C4_PK16_WIDE_DUMPS=0)
set -u
mkdir -p gpurun_out/r3o
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -x -q -m gpu -k "windowed or packed or window_kernel" 2>&1 | tail -3
for rep in 1 2; do
for v in A B; do
lib=exonerate_amd/libc4gpu.so; [ $v = B ] && lib=exonerate_amd/libc4gpu_B.so
C4GPU_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-revcomp --no-cpu-baseline > gpurun_out/r3o/bench_$v$rep.json 2> gpurun_out/r3o/bench_$v$rep.err; echo "bench $v rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3o/bench_$v$rep.json').read().strip().splitlines()[-1])
print("$v$rep ms_per_step %.1f" % d["ms_per_step"], {k: round(v/3,1) for k,v in d.get("kernel_ms",{}).items()})
PY
done
done
for k in 12; do
C4GPU_SEED_KSHIFT=$k python bench.py --steps 3 --warmup 1 --no-revcomp --no-cpu-baseline > gpurun_out/r3o/bench_k$k.json 2> gpurun_out/r3o/bench_k$k.err; echo "bench kshift $k rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3o/bench_k$k.json').read().strip().splitlines()[-1])
print("kshift $k ms_per_step %.1f" % d["ms_per_step"], {k: round(v/3,1) for k,v in d.get("kernel_ms",{}).items()})
PY
done
