#!/bin/bash
# round 4, call g: the default bench line (configs block: C5's score pass on eight waves of one row per lane), printers on device
# alignments, the round's profiles (two lanes and one lane)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4g; mkdir -p $OUT
timeout 1200 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
python - <<P
import json
d=json.load(open("$OUT/bench.json")); print("bench", round(d["ms_per_step"],1), d["value"], {k: round(v,1) for k,v in d["kernel_ms"].items()}, "revcomp", d["revcomp"]["value"])
for k,v in d["configs"].items(): print(k, round(v["ms_per_pass"],1), "%.3e"%v["value"], v["checked"][:30])
P
timeout 900 python -m pytest tests/test_gpu_printers.py -m gpu -x -q > $OUT/pytest_printers.log 2>&1
tail -3 $OUT/pytest_printers.log
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py -m gpu -x -q -k "staged or packed" > $OUT/pytest_kv.log 2>&1
tail -3 $OUT/pytest_kv.log
bash tools/profile_round.sh r04_b > $OUT/prof_b.log 2>&1
cd $ROOT
BENCH_EXTRA="--no-cpu-baseline --no-revcomp --no-configs" C4GPU_LANES=1 bash tools/profile_round.sh r04_b_lanes1 > $OUT/prof_b_lanes1.log 2>&1
cd $ROOT
timeout 900 python -m pytest tests/test_integration_gpu.py -m gpu -x -q -k "ordinary_exit or small_work" > $OUT/pytest_exit.log 2>&1
tail -3 $OUT/pytest_exit.log
