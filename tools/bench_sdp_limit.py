#!/usr/bin/env python3
"""Which HSP-box size should the SDP seam still send to the device?  The default heuristic mode (--gappedextension yes)
of the drop-in on north-star-shaped input (N cDNAs x N windows of 100 kb, all against all, both strands) with
C4GPU_SDP_MAX_CELLS swept; output compared with the reference's byte for byte.  Markdown table on stdout."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exonerate_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
out = "/tmp/sdplimit"
os.makedirs(out, exist_ok=True)
pairs = workloads.est2genome_pairs(n, 1000, 100000, seed=20260928 + 4)
for path, recs in ((out + "/q.fa", [("cdna%d" % i, p[0]) for i, p in enumerate(pairs)]),
                   (out + "/t.fa", [("win%d" % i, p[1]) for i, p in enumerate(pairs)])):
    with open(path, "w") as f:
        for name, s in recs:
            f.write(">%s\n%s\n" % (name, s.decode()))
args = ["-m", "est2genome", "--showalignment", "no", "--showvulgar", "yes", "-V", "0", out + "/q.fa", out + "/t.fa"]


def run(exe, env=None):
    e = dict(os.environ, C4GPU_VERBOSE="1")
    e.update(env or {})
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        times.append(time.perf_counter() - t0)
        assert r.returncode == 0, r.stderr.decode()[-800:]
    return r.stdout, sorted(times)[1], r.stderr.decode()


ref, t_ref, _ = run(ROOT + "/oracle/_ref/exonerate-compiled")
print("| C4GPU_SDP_MAX_CELLS | wall s (median of 3) | pairs served from device batches | ms in the batches |")
print("|---|---|---|---|")
print("| reference (compiled scheduler, 1 core) | %.2f | - | - |" % t_ref)
for lim in ("0", "4e6", "1e7", "2e7", "3e7", "4e7", "6e7", "1e8", "1e12"):
    o, t, err = run(ROOT + "/integration/_build/exonerate-gpu", {"C4GPU_SDP_MAX_CELLS": lim})
    assert o == ref, lim
    m = re.search(r"c4gpu sdp: (\d+) pairs in (\d+) flush\(es\): (\d+) served from device batches \((\d+) alignments\); batches (\d+) ms", err)
    print("| %s | %.2f | %s | %s |" % (lim, t, ("%s of %s" % (m.group(3), m.group(1))) if m else "-", m.group(5) if m else "-"))
print("\nEvery output byte-identical to the reference's (%d alignments)." % ref.count(b"vulgar:"))
