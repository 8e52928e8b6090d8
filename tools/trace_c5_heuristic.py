#!/usr/bin/env python3
"""Where the wall time of BASELINE config 5's heuristic leg goes in the drop-in: 256 proteins (300 aa) against one 10 Mb chromosome,
-m protein2genome, default mode and --gappedextension no, run with C4GPU_TRACE=1 (shim_mark / engine traces on stderr) and
C4GPU_VERBOSE=1.  Writes <out>/<mode>.err and prints the marks (profiles/r04_c5_breakdown.md)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exonerate_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/c5t"
os.makedirs(out, exist_ok=True)
t0 = time.perf_counter()
proteins, contig, places = workloads.protein_vs_contig(n, 300, 10000000, seed=20260935, introns=True)
with open(out + "/q.fa", "w") as f:
    for i, p in enumerate(proteins):
        f.write(">p%d\n%s\n" % (i, p.decode()))
with open(out + "/t.fa", "w") as f:
    f.write(">chr\n%s\n" % contig.decode())
print("inputs written in %.1f s" % (time.perf_counter() - t0))
exe = ROOT + "/integration/_build/exonerate-gpu"
base = ["-m", "protein2genome", "--showalignment", "no", "--showvulgar", "yes", "-V", "0"]
for name, extra in (("default", []), ("bsdp", ["--gappedextension", "no"])):
    for rep in range(2):
        e = dict(os.environ, C4GPU_VERBOSE="1", C4GPU_TRACE="1")
        t0 = time.perf_counter()
        r = subprocess.run([exe] + base + extra + [out + "/q.fa", out + "/t.fa"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr.decode()[-1000:]
    open(out + "/%s.err" % name, "w").write(r.stderr.decode())
    print("== %s: %.2f s wall, %d alignments" % (name, dt, r.stdout.decode().count("vulgar:")))
    marks = [l for l in r.stderr.decode().splitlines() if l.startswith("c4gpu mark:")]
    prev = 0.0
    agg = {}
    for l in marks:
        ms = float(l.split()[2]); what = l.split("ms", 1)[1].strip()
        agg.setdefault(what, [0, 0.0]); agg[what][0] += 1; agg[what][1] += ms - prev
        prev = ms
    for what, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %9.1f ms in %5d interval(s) ending at: %s" % (tot, cnt, what))
    for l in r.stderr.decode().splitlines():
        if "c4gpu " in l and any(k in l for k in ("sdp:", "hsp:", "bsdp:", "seed:", "start-up")):
            print("  " + l.strip()[:400])
