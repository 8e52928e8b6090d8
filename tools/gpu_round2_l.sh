set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sdp.py -m gpu -q --maxfail=10 -p no:cacheprovider -x > gpurun_out/pytest_gpu_l_sdp.log 2>&1
tail -30 gpurun_out/pytest_gpu_l_sdp.log
timeout 1200 python -m pytest tests/test_abi.py tests/test_integration_gpu.py -m gpu -q --maxfail=5 -p no:cacheprovider > gpurun_out/pytest_gpu_l_int.log 2>&1
tail -5 gpurun_out/pytest_gpu_l_int.log
