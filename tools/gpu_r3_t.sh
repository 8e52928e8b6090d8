# round 3: the drop-in's way out (main wrapper, _exit with the device thread started): the short runs of
# test_small_work_does_not_wait_for_the_device repeated, then the integration tests
set -u
mkdir -p gpurun_out/r3t
for rep in 1 2 3 4 5 6 7 8; do
python -m pytest tests/test_integration_gpu.py -x -q -m gpu -k "small_work_does_not_wait" 2>&1 | tail -1
done
python -m pytest tests/test_integration_gpu.py tests/test_integration_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -3
