#!/usr/bin/env python3
"""The sub-optimal loop (exonerate's default --subopt yes) on the north-star batch: first alignments, then one
more round with everything found so far blocked.  Prints a small markdown table."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exonerate_amd as ex
from exonerate_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
eng = ex.Engine(0)
pairs = workloads.est2genome_pairs(n, 1000, 100000)
b = ex.ResidentBatch(eng, ex.Model("est2genome"), pairs)
b.run(2, 32, thr); b.next_paths(32, thr)            # warm-up (and the hit-rate history of the context)
rows = []
for rep in range(2):
    t0 = time.perf_counter(); b.run(2, 32, thr); t1 = time.perf_counter()
    found = b.next_paths(32, thr); t2 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, found))
cells = sum((len(q) + 1) * (len(t) + 1) for q, t in pairs)
print("| threshold | pairs | first alignments ms | second round ms | alignments found in round 2 | cells/s over both rounds |")
print("|---|---|---|---|---|---|")
for a, c, f in rows:
    print("| %d | %d | %.0f | %.0f | %d | %.3g |" % (thr, n, a * 1e3, c * 1e3, f, 2 * cells / (a + c)))
b.close()
