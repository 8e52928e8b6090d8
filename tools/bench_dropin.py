#!/usr/bin/env python3
"""Time the drop-in binary (integration/_build/exonerate-gpu, batching seam on) against the unmodified
reference (oracle/_ref/exonerate-compiled) on north-star-shaped input: NQ cDNAs x NT genomic windows,
all-vs-all exhaustive est2genome.  The reference runs on a sample of the queries only (it needs ~4.5 s per
pair); outputs for the sample are compared byte for byte.  Writes a small markdown table to stdout."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exonerate_amd import workloads

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sample = int(sys.argv[3]) if len(sys.argv) > 3 else 1
out = sys.argv[4] if len(sys.argv) > 4 else "/tmp/dropin"
os.makedirs(out, exist_ok=True)
pairs = workloads.est2genome_pairs(max(nq, nt), 1000, 100000, seed=20260928 + 4)
def fasta(path, recs):
    with open(path, "w") as f:
        for name, s in recs:
            f.write(">%s\n%s\n" % (name, s if isinstance(s, str) else s.decode()))
fasta(out + "/q.fa", [("cdna%d" % i, pairs[i][0]) for i in range(nq)])
fasta(out + "/qs.fa", [("cdna%d" % i, pairs[i][0]) for i in range(max(1, sample))])
fasta(out + "/t.fa", [("win%d" % i, pairs[i][1]) for i in range(nt)])
args = ["-m", "est2genome", "-E", "yes", "-S", "no", "--showalignment", "no", "--showvulgar", "yes", "-V", "0"]
def run(exe, q, env=None):
    e = dict(os.environ); e.update(env or {})
    t0 = time.perf_counter()
    r = subprocess.run([exe] + args + [q, out + "/t.fa"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    return r.stdout.decode(), dt
gpu_exe = ROOT + "/integration/_build/exonerate-gpu"
cpu_exe = ROOT + "/oracle/_ref/exonerate-compiled"
run(gpu_exe, out + "/qs.fa")                                  # warm-up (HIP module load)
g_all, t_gpu = run(gpu_exe, out + "/q.fa")
if sample == 0:                     # timing of the batched binary only (the reference needs ~15 s per pair)
    cells = 1001 * 100001
    print("| run | pairs | wall s | pairs/s | first-pass cells/s |")
    print("|---|---|---|---|---|")
    print("| exonerate-gpu, batching seam | %d | %.2f | %.2f | %.3g |" % (nq * nt, t_gpu, nq * nt / t_gpu, nq * nt * cells / t_gpu))
    print("\nvulgar lines: gpu %d" % g_all.count("vulgar:"))
    sys.exit(0)
g_call, t_call = run(gpu_exe, out + "/qs.fa", {"C4GPU_BATCH": "0"})
g_s, t_gs = run(gpu_exe, out + "/qs.fa")
c_s, t_cpu = run(cpu_exe, out + "/qs.fa")
assert g_s == c_s and g_call == c_s, "outputs differ"
vul = lambda s: [l for l in s.splitlines() if l.startswith("vulgar:")]
assert vul(g_all)[:len(vul(c_s))] == vul(c_s), "the sample's alignments are not the head of the full run"
cells = 1001 * 100001
print("| run | pairs | wall s | pairs/s | first-pass cells/s |")
print("|---|---|---|---|---|")
print("| exonerate-gpu, batching seam | %d | %.2f | %.2f | %.3g |" % (nq * nt, t_gpu, nq * nt / t_gpu, nq * nt * cells / t_gpu))
print("| exonerate-gpu, per-call shim (C4GPU_BATCH=0) | %d | %.2f | %.2f | %.3g |" % (sample * nt, t_call, sample * nt / t_call, sample * nt * cells / t_call))
print("| exonerate-gpu, batching seam, sample | %d | %.2f | %.2f | %.3g |" % (sample * nt, t_gs, sample * nt / t_gs, sample * nt * cells / t_gs))
print("| exonerate (reference, compiled Viterbi, 1 core) | %d | %.2f | %.2f | %.3g |" % (sample * nt, t_cpu, sample * nt / t_cpu, sample * nt * cells / t_cpu))
print("\nvulgar lines: gpu %d, sample %d; sample output byte-identical: yes" % (g_all.count("vulgar:"), c_s.count("vulgar:")))
