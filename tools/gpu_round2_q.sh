set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_sdp.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_q_sdp.log 2>&1
tail -12 gpurun_out/pytest_gpu_q_sdp.log
timeout 2400 python -m pytest tests/test_integration_fuzz_gpu.py tests/test_integration_gpu.py -m gpu -q -p no:cacheprovider -k "heuristic or selenocysteine" > gpurun_out/pytest_gpu_q_int.log 2>&1
tail -12 gpurun_out/pytest_gpu_q_int.log
