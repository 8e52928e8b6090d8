#!/bin/bash
# round 4, call r: the default bench line with the c5_heuristic leg; the pruned GPU suite's time
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4r; mkdir -p $OUT
( time timeout 1200 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
python - <<P
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("bench", round(d["ms_per_step"],1), "%.3e"%d["value"], "revcomp %.3e"%d["revcomp"]["value"], "valu", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["roofline"]["valu"].items() if k in ("frac","achieved","peak")}, d["roofline"]["valu"].get("mix_rate_at_occupancy"))
for k,v in d["configs"].items(): print(" ", k, {kk: vv for kk,vv in v.items() if kk in ("ms_per_pass","value","wall_s","checked")})
P
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
