# round 3: strip carry rows of the checkpoint pass through an LDS stage filled 64 columns ahead (C4_CARRY_STAGE=1): S3 = 3 rows x 2 waves, S2 = 2 rows x 2 waves; A = the tree's library
set -u
mkdir -p gpurun_out/r3s
for v in ${VARIANTS:-A S3 S2 A S3 S2}; do
lib=exonerate_amd/libc4gpu.so; [ $v != A ] && lib=exonerate_amd/libc4gpu_$v.so
C4GPU_LIB=$PWD/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-revcomp --no-cpu-baseline > gpurun_out/r3s/bench_$v.json 2> gpurun_out/r3s/bench_$v.err; echo "bench $v rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3s/bench_$v.json').read().strip().splitlines()[-1])
print("$v ms_per_step %.1f" % d["ms_per_step"], {k: round(v/3,1) for k,v in d.get("kernel_ms",{}).items()}, "aligned_ok", d["config"].get("aligned_in_sample"))
PY
done
for v in S2 S3; do
C4GPU_LIB=$PWD/exonerate_amd/libc4gpu_$v.so timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
done
