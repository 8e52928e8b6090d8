# round 3, first device run of the sparse SDP wavefront: library tests, drop-in tests, the north-star-shaped heuristic run
set -u
mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_sdp.py -x -q > gpurun_out/r3a/pytest_sdp.log 2>&1; echo "sdp tests rc=$?"; tail -5 gpurun_out/r3a/pytest_sdp.log
python -m pytest tests/test_integration_gpu.py -x -q -k "heuristic_sdp" > gpurun_out/r3a/pytest_int.log 2>&1; echo "integration rc=$?"; tail -5 gpurun_out/r3a/pytest_int.log
C4GPU_TRACE=1 python tools/bench_heuristic.py 32 gpurun_out/r3a/heur > gpurun_out/r3a/heuristic.md 2> gpurun_out/r3a/heuristic.err; echo "heur rc=$?"; tail -12 gpurun_out/r3a/heuristic.md; tail -5 gpurun_out/r3a/heuristic.err
