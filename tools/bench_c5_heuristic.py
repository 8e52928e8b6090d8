#!/usr/bin/env python3
"""BASELINE config 5's heuristic leg: N proteins (300 aa) against one 10 Mb chromosome with their intron-split genes,
-m protein2genome, default mode (seeding + SDP) and --gappedextension no (BSDP): the reference binary (1 core) against the
drop-in; outputs compared byte for byte.  Markdown table on stdout."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exonerate_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/c5h"
os.makedirs(out, exist_ok=True)
proteins, contig, places = workloads.protein_vs_contig(n, 300, 10000000, seed=20260935, introns=True)
with open(out + "/q.fa", "w") as f:
    for i, p in enumerate(proteins):
        f.write(">p%d\n%s\n" % (i, p.decode()))
with open(out + "/t.fa", "w") as f:
    f.write(">chr\n%s\n" % contig.decode())
gpu_exe = ROOT + "/integration/_build/exonerate-gpu"
cpu_exe = ROOT + "/oracle/_ref/exonerate-compiled"
base = ["-m", "protein2genome", "--showalignment", "no", "--showvulgar", "yes", "-V", "0"]


def run(exe, extra, env=None):
    e = dict(os.environ, C4GPU_VERBOSE="1")
    e.update(env or {})
    t0 = time.perf_counter()
    r = subprocess.run([exe] + base + extra + [out + "/q.fa", out + "/t.fa"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    return r.stdout.decode(), dt, r.stderr.decode()


print("# C5 heuristic leg: %d proteins x one 10 Mb chromosome, -m protein2genome\n" % n)
print("| mode | reference wall s (1 core) | exonerate-gpu wall s | alignments | device work |")
print("|---|---|---|---|---|")
for name, extra in (("default (SDP)", []), ("--gappedextension no (BSDP)", ["--gappedextension", "no"])):
    ref, t_ref, _ = run(cpu_exe, extra)
    gpu, t_gpu, err = run(gpu_exe, extra)
    assert gpu == ref, "outputs differ in mode %s" % name
    notes = "; ".join(l.split("c4gpu ", 1)[1].strip() for l in err.splitlines() if "c4gpu " in l and ("sdp:" in l or "hsp:" in l or "bsdp:" in l))
    print("| %s | %.1f | %.1f | %d | %s |" % (name, t_ref, t_gpu, ref.count("vulgar:"), notes))
print("\nOutputs byte-identical.")
