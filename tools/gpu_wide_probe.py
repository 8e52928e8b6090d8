"""The wide-region route: 64 cDNAs x 64 windows all against all (4 096 rectangles; 64 real alignments, 4 032 chance alignments
that span most of their window -- what reverse strands and all-against-all runs are made of).  Kernel time per pass, HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exonerate_amd as ex
from exonerate_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
base = workloads.est2genome_pairs(n, 1000, 100000)
pairs = [(base[i][0], base[j][1]) for i in range(n) for j in range(n)]
eng = ex.Engine(0); model = ex.Model("est2genome")
b = ex.ResidentBatch(eng, model, pairs)
b.kernel_stats(0, reset=True)
names = {0: "score", 1: "path", 2: "region", 3: "checkpoint"}
for rep in range(3):
    for m in range(4):
        b.kernel_stats(m, reset=True)
    if rep == 2 and os.environ.get("PROBE_TRACE"):
        os.environ["C4GPU_TRACE"] = "1"
    t0 = time.perf_counter()
    b.run(2, threshold=int(os.environ.get("PROBE_THRESHOLD", "100")))
    dt = (time.perf_counter() - t0) * 1e3
    ks = {names[m]: b.kernel_stats(m) for m in range(4)}
    print("run %d: %.1f ms  " % (rep, dt) + "  ".join("%s %.1f ms / %d" % (k, v["ms"], v["launches"]) for k, v in ks.items()), flush=True)
sc, rg = b.scores()
import statistics
w = [r[3] for r in rg]
print("scores: median %d; region widths: median %d, mean %.0f; valid %d" % (statistics.median(sc), statistics.median(w), sum(w) / len(w),
      sum(1 for i in range(len(pairs)) if b.alignment(i) is not None) if len(pairs) <= 4096 else -1))
b.close(); eng.close()
