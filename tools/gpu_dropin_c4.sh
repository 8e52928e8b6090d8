#!/bin/bash
# BASELINE config 4 through the drop-in binary (64 x 64 all against all), with the shim's marks; extra environment in $DROPIN_ENV
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/c4d && python - <<'PY'
import sys; sys.path.insert(0,'.')
from exonerate_amd import workloads
workloads.write_c4_dropin_input('/tmp/c4d')
PY
ARGS="-m est2genome -E yes -S no --revcomp no --showalignment no --showvulgar yes -V 0"
for rep in 1 2 3; do
  env C4GPU_VERBOSE=1 C4GPU_TRACE=1 $DROPIN_ENV integration/_build/exonerate-gpu $ARGS /tmp/c4d/q.fa /tmp/c4d/t.fa > /tmp/c4d/out.$rep 2> /tmp/c4d/err.$rep
  echo "rc $? rep $rep env [$DROPIN_ENV]"
  grep -E "c4gpu mark|flush of|start-up|find_path_batch: alignments assembled|staging: coded" /tmp/c4d/err.$rep | cut -c1-330
  sha256sum /tmp/c4d/out.$rep; grep -c vulgar /tmp/c4d/out.$rep
done
cp /tmp/c4d/err.3 gpurun_out/dropin_c4_err.log
