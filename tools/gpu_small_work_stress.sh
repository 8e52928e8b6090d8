#!/bin/bash
# the drop-in on small heuristic work (tests/test_integration_gpu.py::test_small_work_does_not_wait_for_the_device, first case), N
# times, alone and beside a process that keeps taking and giving back device memory: exit status and output of every run
cd $GRAFT_REPO_ROOT
N=${1:-150}
mkdir -p /tmp/sw && python - <<'PY'
import sys; sys.path.insert(0, '.')
from exonerate_amd import workloads
pairs = workloads.est2genome_pairs(8, 1000, 100000, seed=77)
open('/tmp/sw/q.fa', 'w').write("".join(">q%d\n%s\n" % (k, q.decode()) for k, (q, t) in enumerate(pairs)))
open('/tmp/sw/t.fa', 'w').write("".join(">t%d\n%s\n" % (k, t.decode()) for k, (q, t) in enumerate(pairs)))
PY
ARGS="-m est2genome --gappedextension no -S no --showalignment yes --showvulgar yes -V 0 /tmp/sw/q.fa /tmp/sw/t.fa"
oracle/_ref/exonerate-compiled $ARGS > /tmp/sw/ref.out 2>/dev/null
export C4GPU_SEGV_TRACE=1
[ -n "$STRESS_ENV" ] && export $STRESS_ENV       # e.g. STRESS_ENV=C4GPU_FAST_EXIT=0: the ordinary exit
run_loop() {
  local bad=0
  for i in $(seq 1 $N); do
    integration/_build/exonerate-gpu $ARGS > /tmp/sw/out 2> /tmp/sw/err; rc=$?
    if [ $rc -ne 0 ] || ! cmp -s /tmp/sw/out /tmp/sw/ref.out; then
      bad=$((bad+1)); echo "== $1 run $i rc $rc"; head -60 /tmp/sw/err | cut -c1-200
    fi
  done
  echo "$1: $bad bad of $N"
}
run_loop alone
python - <<'PY' &
import torch, time
t0 = time.time()
while time.time() - t0 < 120:
    x = torch.empty(100 << 30, dtype=torch.uint8, device='cuda'); x.zero_(); torch.cuda.synchronize(); del x; torch.cuda.empty_cache(); time.sleep(0.05)
PY
HOG=$!
sleep 8
run_loop beside-a-memory-hog
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
