set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernel_variants.py tests/test_library_fuzz_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "windowed or fuzz" > gpurun_out/pytest_gpu_c1.log 2>&1
tail -25 gpurun_out/pytest_gpu_c1.log
(cd /tmp && C4GPU_TRACE=1 timeout 900 python $ROOT/bench.py --steps 3 --warmup 1 --no-revcomp > $ROOT/gpurun_out/bench_r02_c.json 2> $ROOT/gpurun_out/bench_r02_c.err)
tail -c 1500 gpurun_out/bench_r02_c.json
grep -E "windowed|find_path_batch" gpurun_out/bench_r02_c.err | tail -12
(cd /tmp && C4GPU_WINDOWED=0 timeout 900 python $ROOT/bench.py --steps 3 --warmup 1 --no-revcomp --no-cpu-baseline > $ROOT/gpurun_out/bench_r02_c_onepass.json 2> $ROOT/gpurun_out/bench_r02_c_onepass.err)
tail -c 600 gpurun_out/bench_r02_c_onepass.json
