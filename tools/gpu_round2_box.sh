set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_sdp.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_box.log 2>&1
tail -4 gpurun_out/pytest_gpu_box.log; grep -n "AssertionError:" gpurun_out/pytest_gpu_box.log | head -5
C4GPU_SDP_MARGIN=8 timeout 1200 python -X faulthandler -m pytest tests/test_gpu_sdp.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_box8.log 2>&1
tail -3 gpurun_out/pytest_gpu_box8.log; grep -n "AssertionError:" gpurun_out/pytest_gpu_box8.log | head -5
timeout 1800 python -m pytest tests/test_integration_gpu.py tests/test_integration_fuzz_gpu.py -m gpu -q -p no:cacheprovider -k "sdp or c1 or heuristic" > gpurun_out/pytest_gpu_box_int.log 2>&1
tail -3 gpurun_out/pytest_gpu_box_int.log
(cd /tmp && timeout 600 python $ROOT/tools/bench_sdp.py 100 > $ROOT/gpurun_out/sdp_bench.md 2> $ROOT/gpurun_out/sdp_bench.err)
head -4 gpurun_out/sdp_bench.md | tail -2
(cd /tmp && timeout 1500 python $ROOT/tools/bench_heuristic.py 32 > $ROOT/gpurun_out/heuristic_box.md 2> $ROOT/gpurun_out/heuristic_box.err)
tail -12 gpurun_out/heuristic_box.md; tail -3 gpurun_out/heuristic_box.err
