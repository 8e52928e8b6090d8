set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_hsp.py tests/test_integration_gpu.py tests/test_integration_fuzz_gpu.py tests/test_abi.py -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu_j.log 2>&1
tail -15 gpurun_out/pytest_gpu_j.log
(cd /tmp && timeout 600 python $ROOT/tools/bench_hsp.py 64 > $ROOT/gpurun_out/hsp_j.md 2> $ROOT/gpurun_out/hsp_j.err)
cat gpurun_out/hsp_j.md; tail -3 gpurun_out/hsp_j.err
(cd /tmp && timeout 1500 python $ROOT/tools/bench_heuristic.py 32 > $ROOT/gpurun_out/heuristic_j.md 2> $ROOT/gpurun_out/heuristic_j.err)
cat gpurun_out/heuristic_j.md; tail -3 gpurun_out/heuristic_j.err
