#!/bin/bash
# round 5, final library: the full bench line, then the profiles of tools/profile_round.sh on two lanes and on one
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_e_bench_full.json 2> gpurun_out/r05_e_bench_full.err
BENCH_EXTRA="--steps 6 --warmup 3 --no-configs --no-revcomp" bash tools/profile_round.sh r05_e > gpurun_out/prof_r05_e.log 2>&1
C4GPU_LANES=1 BENCH_EXTRA="--steps 6 --warmup 3 --no-configs --no-revcomp" bash tools/profile_round.sh r05_e_lanes1 > gpurun_out/prof_r05_e_lanes1.log 2>&1
ls gpurun_out/prof_r05_e gpurun_out/prof_r05_e_lanes1
