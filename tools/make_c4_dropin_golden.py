#!/usr/bin/env python3
"""tests/golden/bench_c4_dropin.json: what the reference binary (oracle/_ref/exonerate-compiled) prints for BASELINE config 4
through the command line -- 64 cDNAs of 1 kb against 64 genomic windows of 100 kb, all against all = 4 096 rectangles,
-m est2genome -E yes -S no --revcomp no -- as a SHA-256 of its stdout.  The reference needs ~4.6 s per pair on one core (5 h for
the whole input), so it runs one process per query (exonerate's own --querychunkid / --querychunktotal, exonerate.c:64-75:
chunk k of 64 = query k) on several cores, and the chunks' outputs are concatenated in query order, which is the order a single
process prints them in (the query loop is the outer one, fastapipe.c).  bench.py's `configs.c4_dropin` runs the drop-in on the
same files and compares.  usage: make_c4_dropin_golden.py [workdir] [processes] [queries]"""
import hashlib, json, os, subprocess, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exonerate_amd import workloads
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/c4_dropin_golden"
procs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.makedirs(out, exist_ok=True)
NQ = NT = 64
ARGS = ["-m", "est2genome", "-E", "yes", "-S", "no", "--revcomp", "no", "--showalignment", "no", "--showvulgar", "yes", "-V", "0"]
qf, tf = workloads.write_c4_dropin_input(out, NQ, NT)
exe = os.path.join(ROOT, "oracle", "_ref", "exonerate-compiled")


def chunk(k):
    path = os.path.join(out, "chunk%02d.out" % k)
    if os.path.exists(path):                 # resumable: a finished chunk is kept
        return open(path, "rb").read()
    r = subprocess.run([exe] + ARGS + ["--querychunkid", str(k + 1), "--querychunktotal", str(NQ), qf, tf],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    open(path + ".tmp", "wb").write(r.stdout)
    os.rename(path + ".tmp", path)
    return r.stdout


# the reference needs ~30 s per chance alignment across a 100 kb window: `upto` queries only (default 5 = 320 rectangles, ~32 min
# each on one core); chunks already on disk are kept, so the golden file can be extended later
upto = int(sys.argv[3]) if len(sys.argv) > 3 else 5
t0 = time.perf_counter()
with ThreadPoolExecutor(procs) as pool:
    parts = list(pool.map(chunk, range(upto)))
dt = time.perf_counter() - t0
text = b"".join(parts)
json.dump({"args": ARGS, "queries": NQ, "targets": NT, "checked_queries": upto, "sha256_head": hashlib.sha256(text).hexdigest(),
           "head_bytes": len(text), "head_alignments": text.decode().count("vulgar:"), "first_lines": text.decode().splitlines()[:2],
           "reference_s_per_query_one_core_build_container": round(dt * min(procs, upto) / max(1, upto)),
           "note": "the first %d queries (= %d rectangles) through the reference binary, one process per query (--querychunkid k "
                   "--querychunktotal 64); the drop-in's stdout must begin with exactly their output (the query loop is the outer one)"
                   % (upto, upto * NT)},
          open(os.path.join(ROOT, "tests", "golden", "bench_c4_dropin.json"), "w"), indent=1)
print("reference: %.1f s on %d processes, %d alignments of the first %d queries" % (dt, procs, text.decode().count("vulgar:"), upto))
