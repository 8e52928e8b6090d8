#!/bin/bash
# The library a round ends on: whole GPU suite, smoke, the default bench line, profiles (two lanes, one lane, SQ breakdown).
# usage (through gpurun, from the repo root): bash tools/gpu_round_end.sh r04_e   -> gpurun_out/prof_<tag>*, gpurun_out/<tag>/
set -u
TAG=${1:-r04_e}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 1800 python -m pytest tests/ -x -q -m gpu --durations=15 > $OUT/pytest_gpu.log 2>&1
tail -22 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/profile_round.sh $TAG > $OUT/prof.log 2>&1
cd $ROOT
BENCH_EXTRA="--no-cpu-baseline --no-revcomp --no-configs" C4GPU_LANES=1 bash tools/profile_round.sh ${TAG}_lanes1 > $OUT/prof_lanes1.log 2>&1
cd $ROOT
C4GPU_LANES=1 bash tools/profile_sq_breakdown.sh $TAG > $OUT/sq.log 2>&1
cd $ROOT
python - <<P
import json
for t in ("$TAG","${TAG}_lanes1"):
    try:
        d=json.loads(open("gpurun_out/prof_%s/bench.json"%t).read().strip().splitlines()[-1]); print(t, round(d["ms_per_step"],1), "%.3e"%d["value"], {k: round(v,1) for k,v in d["kernel_ms"].items()}, "revcomp", (d.get("revcomp") or {}).get("value"))
        for k,v in (d.get("configs") or {}).items(): print("  ",k, round(v.get("ms_per_pass",0),1), v.get("value"), v.get("wall_s"))
    except Exception as e: print(t,"failed",e)
P
