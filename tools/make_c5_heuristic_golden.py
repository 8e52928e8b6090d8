#!/usr/bin/env python3
"""tests/golden/bench_c5_heuristic.json: what the reference binary (oracle/_ref/exonerate-compiled, one core, ~70 s) prints for
BASELINE config 5's heuristic leg -- 256 proteins of 300 aa against one 10 Mb chromosome, -m protein2genome, default mode -- as
a SHA-256 of its stdout and its number of alignments; bench.py's `configs.c5_heuristic` runs the drop-in on the same input and
compares (the reference itself is too slow for the bench)."""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exonerate_amd import workloads
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/c5h_golden"
os.makedirs(out, exist_ok=True)
ARGS = ["-m", "protein2genome", "--showalignment", "no", "--showvulgar", "yes", "-V", "0"]
qf, tf = workloads.write_c5_heuristic_input(out)
t0 = time.perf_counter()
r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "exonerate-compiled")] + ARGS + [qf, tf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
dt = time.perf_counter() - t0
assert r.returncode == 0, r.stderr.decode()[-1000:]
json.dump({"args": ARGS, "sha256": hashlib.sha256(r.stdout).hexdigest(), "alignments": r.stdout.decode().count("vulgar:"),
           "reference_wall_s_one_core_build_container": round(dt, 1)},
          open(os.path.join(ROOT, "tests", "golden", "bench_c5_heuristic.json"), "w"), indent=1)
print("reference: %.1f s, %d alignments" % (dt, r.stdout.decode().count("vulgar:")))
