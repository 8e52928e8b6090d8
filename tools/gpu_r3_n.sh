# round 3: checkpoint pass with arithmetic winner masks (C4_ARITH_SELECT=1 build of k_est2genome_ckpt_cont_local in libc4gpu_B.so) against the default
set -u
mkdir -p gpurun_out/r3n
for rep in 1 2; do
for v in A B; do
lib=exonerate_amd/libc4gpu.so; [ $v = B ] && lib=exonerate_amd/libc4gpu_B.so
C4GPU_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-revcomp --no-cpu-baseline > gpurun_out/r3n/bench_$v$rep.json 2> gpurun_out/r3n/bench_$v$rep.err; echo "bench $v rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3n/bench_$v$rep.json').read().strip().splitlines()[-1])
print("$v$rep ms_per_step %.1f" % d["ms_per_step"], {k: round(v/3,1) for k,v in d.get("kernel_ms",{}).items()})
PY
done
done
C4GPU_LIB=$PWD/exonerate_amd/libc4gpu_B.so timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
