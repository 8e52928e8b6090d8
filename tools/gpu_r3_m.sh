set -u
mkdir -p gpurun_out/r3m
python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q > gpurun_out/r3m/pytest.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r3m/pytest.log
for pk in 1 0; do C4GPU_PK16=$pk python tools/bench_configs.py c2 c3 > gpurun_out/r3m/configs_pk$pk.md 2> gpurun_out/r3m/configs_pk$pk.err; echo "configs pk=$pk rc=$?"; cat gpurun_out/r3m/configs_pk$pk.md; done
