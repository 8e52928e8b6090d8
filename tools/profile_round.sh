#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ activity) as the microarch guide prescribes
# Everything lands under gpurun_out/prof_<tag>/; tools/summarise_profile.py turns it into profiles/.
set -u
TAG=${1:-r02_a}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs"
(cd $ROOT && python -c "from exonerate_amd.srchash import csrc_hash; print(csrc_hash())") > "$OUT/csrc_hash.txt"
cd /tmp
# (C4GPU_LANES=1 in the environment: the whole round on one launch lane, so that per-kernel times and counters add up to the step)
python $ROOT/bench.py --steps 2 --warmup 1 ${BENCH_EXTRA:-} > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python $ROOT/bench.py --no-revcomp --no-configs --no-cpu-baseline > "$OUT/trace.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- $BENCH > "$OUT/pmc_$N.log" 2>&1
done
find "$OUT" -name '*.csv' -size +8M -delete
ls -R "$OUT" | head -50
