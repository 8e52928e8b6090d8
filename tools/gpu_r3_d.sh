# round 3: FIND_CHECKPOINTS on cooperating waves (viterbi_kernel_mwc): parity, then the bench with 1 / 4 / 8 waves per job
set -u
mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_parity.py tests/test_library_fuzz_gpu.py -x -q > gpurun_out/r3d/pytest.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/r3d/pytest.log
for w in 0 4 8; do
C4GPU_MWC=$w python bench.py --steps 4 --warmup 1 --no-revcomp --no-cpu-baseline > gpurun_out/r3d/bench_mwc$w.json 2> gpurun_out/r3d/bench_mwc$w.err; echo "bench mwc=$w rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r3d/bench_mwc$w.json').read().strip().splitlines()[-1])
print("mwc=$w ms_per_step %.1f" % d["ms_per_step"], {k: round(v/4,1) for k,v in d["kernel_ms"].items()})
PY
done
