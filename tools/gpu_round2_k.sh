set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
# host loops of the engine split over threads: the library fuzz (dpmemory 0 / 1: nested checkpoint routes), the parity suite, C2
timeout 2400 python -m pytest tests/test_library_fuzz_gpu.py tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/pytest_gpu_k.log 2>&1
tail -8 gpurun_out/pytest_gpu_k.log
(cd /tmp && C4GPU_TRACE=1 timeout 600 python $ROOT/tools/bench_configs.py c2 > $ROOT/gpurun_out/c2_trace_k.md 2> $ROOT/gpurun_out/c2_trace_k.err)
cat gpurun_out/c2_trace_k.md; grep "find_path_batch\|run mode" gpurun_out/c2_trace_k.err | tail -14
(cd /tmp && C4GPU_HOST_THREADS=1 timeout 600 python $ROOT/tools/bench_configs.py c2 > $ROOT/gpurun_out/c2_1thread_k.md 2> /dev/null)
cat gpurun_out/c2_1thread_k.md
(cd /tmp && timeout 900 python $ROOT/bench.py --steps 3 --warmup 1 --no-revcomp > $ROOT/gpurun_out/bench_k.json 2> $ROOT/gpurun_out/bench_k.err)
tail -c 700 gpurun_out/bench_k.json
