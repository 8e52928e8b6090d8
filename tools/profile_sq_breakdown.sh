#!/bin/bash
# Issue-slot breakdown of the step's kernels: SQ instruction-class counters in separate --pmc passes (never combined with a
# trace), written under gpurun_out/prof_<tag>_sq/; tools/summarise_sq.py turns them into profiles/<tag>_sq.csv.
set -u
TAG=${1:-r03_f}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_sq
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-revcomp --no-configs"
cd /tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_IFETCH SQ_IFETCH SQ_INST_LEVEL_VMEM" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_WAVE32"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --output-format csv -d "$OUT/pass$i" -- $BENCH > "$OUT/pass$i.log" 2>&1
  tail -2 "$OUT/pass$i.log" | cut -c1-200
done
find "$OUT" -name '*.csv' -size +8M -delete
ls "$OUT"
