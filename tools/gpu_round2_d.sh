set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/dbg_fuzz4.py gpu > gpurun_out/dbg_fuzz4.log 2>&1
cat gpurun_out/dbg_fuzz4.log | cut -c1-400
timeout 1500 python -m pytest tests/test_gpu_kernel_variants.py tests/test_library_fuzz_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "windowed or fuzz" > gpurun_out/pytest_gpu_d1.log 2>&1
tail -15 gpurun_out/pytest_gpu_d1.log
