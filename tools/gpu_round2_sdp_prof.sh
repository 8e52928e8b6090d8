set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_sdp
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/tools/bench_sdp.py 100 > $OUT/trace.log 2>&1
find $OUT -name '*kernel_stats.csv' | head -2
cat $(find $OUT -name '*kernel_stats.csv' | head -1) | cut -c1-220 | head -12
