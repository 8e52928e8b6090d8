set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { (cd /tmp && env "$@" timeout 600 python $ROOT/bench.py --steps 3 --warmup 1 --no-revcomp --no-cpu-baseline > $ROOT/gpurun_out/bench_h.json 2> $ROOT/gpurun_out/bench_h.err); python -c "
import json,sys;d=json.loads(open('gpurun_out/bench_h.json').read().strip().splitlines()[-1]);print(sys.argv[1:],'%.4g'%d['value'],'%.1f'%d['ms_per_step'],{k:round(v/3,1) for k,v in d['kernel_ms'].items()})" "$@"; }
run A=default
run C4GPU_WPE=1
timeout 2000 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_parity.py tests/test_library_fuzz_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/pytest_gpu_h.log 2>&1
tail -4 gpurun_out/pytest_gpu_h.log
