#!/bin/bash
# the 512-pair shard of a strong-scaled north-star step on one GPU: kernel shapes of the region windows / checkpoint pass, lanes
cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --pairs 512 --no-configs --no-cpu-baseline --no-revcomp --steps 6 --warmup 3 > /tmp/s.json 2> /tmp/s.err
  python -c "
import json;d=json.load(open('/tmp/s.json'));k=d['kernel_ms'];print('%-28s %7.1f ms/step  %.3e cells/s  score %.1f region %.1f ckpt %.1f path %.1f (per step)' % ('$label', d['ms_per_step'], d['value'], k['score']/6, k['region']/6, k['checkpoint']/6, k['path']/6))"
}
run default A=1
run lanes2 C4GPU_LANES=2
for w in 5 6 7 10; do run win16=$w C4GPU_WIN16=$w; done
for c in 5 6 2; do run ck16=$c C4GPU_CK16=$c; done
