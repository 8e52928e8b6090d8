bash tools/gpu_r3_h.sh 2>&1 | grep "sdp round\|wall"
python tools/bench_heuristic.py 32 /tmp/heur > /dev/null 2>&1
C4GPU_TRACE=1 integration/_build/exonerate-gpu -m est2genome --gappedextension yes --showalignment no -V 0 /tmp/heur/q.fa /tmp/heur/t.fa 2>&1 >/dev/null | grep "sdp round"
