# round 3: the word scan on the device (library + drop-in with the seed-for-seed check), start-up timings
set -u
mkdir -p gpurun_out/r3g
python -m pytest tests/test_gpu_seed.py -x -q -s > gpurun_out/r3g/pytest_seed.log 2>&1; echo "seed rc=$?"; tail -6 gpurun_out/r3g/pytest_seed.log
python -m pytest tests/test_integration_gpu.py -x -q -k "word_scan or c5_heuristic or heuristic" > gpurun_out/r3g/pytest_int.log 2>&1; echo "int rc=$?"; tail -6 gpurun_out/r3g/pytest_int.log
python tools/bench_startup.py > gpurun_out/r3g/startup.txt 2> gpurun_out/r3g/startup.err; echo "startup rc=$?"; cat gpurun_out/r3g/startup.txt; tail -3 gpurun_out/r3g/startup.err
python tools/bench_c5_heuristic.py 64 gpurun_out/r3g/c5h > gpurun_out/r3g/c5.md 2>&1; cat gpurun_out/r3g/c5.md | tail -6; rm -rf gpurun_out/r3g/c5h
