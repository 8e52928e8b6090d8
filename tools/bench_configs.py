#!/usr/bin/env python3
"""End-to-end Optimal_find_path throughput of BASELINE.json's other configurations on one MI355X (the
bench line is C4; these are informative).  Prints a markdown table."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exonerate_amd as ex
from exonerate_amd import workloads

eng = ex.Engine(0)
rows = []

def run(name, model, pairs, reps=2):
    batch = ex.ResidentBatch(eng, model, pairs)
    batch.run(2, 32)                                 # warm-up
    for m in range(4):
        batch.kernel_stats(m, reset=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        batch.run(2, 32)
    dt = (time.perf_counter() - t0) / reps
    cells = sum((len(q) + 1) * (len(t) + 1) for q, t in pairs)
    ks = {m: batch.kernel_stats(m) for m in range(4)}
    n_aln = sum(1 for i in range(min(len(pairs), 32)) if batch.alignment(i) is not None)
    batch.close()
    rows.append("| %s | %d | %.3g | %.1f | %.3g | %.0f / %.0f / %.0f / %.0f | %d/%d |" % (
        name, len(pairs), cells, dt * 1e3, cells / dt,
        ks[0]["ms"] / reps, ks[2]["ms"] / reps, ks[3]["ms"] / reps, ks[1]["ms"] / reps, n_aln, min(len(pairs), 32)))

which = sys.argv[1:] or ["c2", "c3", "c5"]
if "c2" in which:
    run("C2 affine:local DNA 1 kb x 1 kb", ex.Model("affine:local"), workloads.affine_dna_pairs(4096, 1000))
if "c3" in which:
    prot, contig, _ = workloads.protein_vs_contig(1024, 500, 1000000 + 1024 * 1600)
    run("C3 protein2dna 500 aa x one 2.6 Mb contig (1 024 planted genes)", ex.Model("protein2dna"), [(p, contig) for p in prot])
if "c5" in which:
    prot, contig, _ = workloads.protein_vs_contig(256, 300, 10000000, seed=20260935, introns=True)
    run("C5-shaped protein2genome 300 aa x one 10 Mb contig, exhaustive", ex.Model("protein2genome"), [(p, contig) for p in prot], reps=1)
print("| config | pairs | first-pass cells | ms / pass | cells/s end-to-end | kernel ms score/region/ckpt/path | aligned (sample) |")
print("|---|---|---|---|---|---|---|")
print("\n".join(rows))
eng.close()
