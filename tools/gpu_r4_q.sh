#!/bin/bash
# round 4, call q: where the reverse strands' time goes (kernel trace of tools/trace_revcomp.py)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4q; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 python $ROOT/tools/trace_revcomp.py > $OUT/rev.txt 2> $OUT/rev.err
tail -3 $OUT/rev.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/tools/trace_revcomp.py > $OUT/trace.log 2>&1
python - <<P
import csv,glob
f=glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:90], r["Calls"], "%.1f ms total"%(float(r["TotalDurationNs"])/1e6), "%.1f avg"%(float(r["AverageNs"])/1e6))
P
find $OUT -name '*.csv' -size +4M -delete
