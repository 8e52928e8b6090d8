#!/usr/bin/env python3
"""Where the reverse strands of the both-strands run go: 1 024 reverse-complemented cDNAs against their windows, C4GPU_TRACE laps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["C4GPU_TRACE"] = "1"
import exonerate_amd as ex
from exonerate_amd import workloads
COMP = bytes.maketrans(b"ACGTN", b"TGCAN")
pairs = workloads.est2genome_pairs(1024, 1000, 100000)
rev = [(q.translate(COMP)[::-1], t) for q, t in pairs]
eng = ex.Engine(0)
b = ex.ResidentBatch(eng, ex.Model("est2genome"), rev)
b.run(2, 32)
sys.stderr.write("==== timed run\n")
t0 = time.perf_counter()
b.run(2, 32)
print("reverse strands, 1024 pairs: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
sc, reg = b.scores()
import numpy as np
ql = np.array([r[2] for r in reg]); tl = np.array([r[3] for r in reg])
print("scores: median %d; region query length median %d, target length median %d, max %d" % (np.median(sc), np.median(ql), np.median(tl), tl.max()))
