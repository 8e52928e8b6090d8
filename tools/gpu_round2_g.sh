set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { (cd /tmp && env "$@" timeout 600 python $ROOT/bench.py --steps 3 --warmup 1 --no-revcomp --no-cpu-baseline > $ROOT/gpurun_out/bench_g.json 2> $ROOT/gpurun_out/bench_g.err); python -c "
import json,sys;d=json.loads(open('gpurun_out/bench_g.json').read().strip().splitlines()[-1]);print(sys.argv[1:],'%.4g'%d['value'],'%.1f'%d['ms_per_step'],{k:round(v/3,1) for k,v in d['kernel_ms'].items()})" "$@"; }
run A=default
run C4GPU_SEED_KSHIFT=13
run C4GPU_SEED_KSHIFT=11
run C4GPU_WINDOWED=0
timeout 1500 python -m pytest tests/test_gpu_kernel_variants.py tests/test_library_fuzz_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "windowed or fuzz or full_size" > gpurun_out/pytest_gpu_g.log 2>&1
tail -5 gpurun_out/pytest_gpu_g.log
(cd /tmp && C4GPU_TRACE=1 timeout 600 python $ROOT/tools/bench_configs.py c2 > $ROOT/gpurun_out/c2_trace.md 2> $ROOT/gpurun_out/c2_trace.err)
cat gpurun_out/c2_trace.md; grep "find_path_batch" gpurun_out/c2_trace.err | tail -12; grep -E "^c4gpu trace: run|at " gpurun_out/c2_trace.err | tail -24
