# round 3: the whole GPU suite with durations
set -u
mkdir -p gpurun_out/r3j
python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r3j/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -40 gpurun_out/r3j/pytest_gpu.log
