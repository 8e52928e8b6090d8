set -u
mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests/test_gpu_kernel_variants.py -x -q -m gpu -k "packed_16_bit or device_route or lanes" 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_library_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -5
