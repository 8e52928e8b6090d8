# round 3: BASELINE's configurations at full size in the suite (C2 4 096 pairs, C3 1 024 proteins, C4 4 096 pairs against the
# reference binary, C5's heuristic leg through the drop-in), with timings
set -u
mkdir -p gpurun_out/r3b
nproc
python -m pytest tests/test_gpu_configs.py -x -q --durations=10 > gpurun_out/r3b/pytest_configs.log 2>&1; echo "configs rc=$?"; tail -18 gpurun_out/r3b/pytest_configs.log
python -m pytest tests/test_integration_gpu.py -x -q -k "c5_heuristic" --durations=5 > gpurun_out/r3b/pytest_c5.log 2>&1; echo "c5 rc=$?"; tail -30 gpurun_out/r3b/pytest_c5.log
