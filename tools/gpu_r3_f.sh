# round 3 profiles: (1) the default bench under rocprofv3 (kernel trace + three PMC passes, tools/profile_round.sh);
# (2) the SDP wavefront kernels inside the drop-in on the north-star-shaped heuristic input (kernel trace + SQ / traffic counters);
# (3) BASELINE config 5's heuristic leg at 256 proteins x 10 Mb, reference against drop-in
set -u
ROOT=$(pwd)
bash tools/profile_round.sh r03_a > gpurun_out/prof_r03_a.log 2>&1; echo "profile_round rc=$?"
OUT=$ROOT/gpurun_out/prof_r03_sdp
mkdir -p $OUT
python tools/bench_heuristic.py 32 $OUT/heur > $OUT/heuristic.md 2> $OUT/heuristic.err; echo "heuristic rc=$?"
export TMPDIR=/tmp
cd /tmp
ARGS="-m est2genome --gappedextension yes --showalignment no --showvulgar yes -V 0 $OUT/heur/q.fa $OUT/heur/t.fa"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $ROOT/integration/_build/exonerate-gpu $ARGS > $OUT/trace.out 2> $OUT/trace.err
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-20)
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- $ROOT/integration/_build/exonerate-gpu $ARGS > $OUT/pmc_$N.out 2> $OUT/pmc_$N.err
done
cd $ROOT
find $OUT -name '*.csv' -size +8M -delete
python tools/bench_c5_heuristic.py 256 $OUT/c5h > $OUT/c5_heuristic.md 2> $OUT/c5_heuristic.err; echo "c5 rc=$?"; cat $OUT/c5_heuristic.md
rm -rf $OUT/c5h $OUT/heur
tail -12 $OUT/heuristic.md
