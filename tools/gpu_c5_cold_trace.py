"""Where a COLD run of config 5's heuristic leg spends its wall time (cold-start attribution): the first
exonerate-gpu process on a fresh box (binary and libraries not in the page cache, no code object loaded yet) with the shim's
marks (C4GPU_TRACE), then two warm ones with the same marks.  Writes gpurun_out/c5_cold_trace.log."""
import hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exonerate_amd import workloads

exe = os.path.join(ROOT, "integration", "_build", "exonerate-gpu")
want = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_c5_heuristic.json")))
out = open(os.path.join(ROOT, "gpurun_out", "c5_cold_trace.log"), "w")
with tempfile.TemporaryDirectory() as d:
    qf, tf = workloads.write_c5_heuristic_input(d)
    env = dict(os.environ, C4GPU_VERBOSE="1", C4GPU_TRACE="1")
    for k in range(3):
        t0 = time.perf_counter()
        r = subprocess.run([exe] + want["args"] + [qf, tf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        dt = time.perf_counter() - t0
        ok = r.returncode == 0 and hashlib.sha256(r.stdout).hexdigest() == want["sha256"]
        out.write("==== run %d (%s): %.3f s wall, output %s\n" % (k, "cold" if k == 0 else "warm", dt, "equal to the reference's" if ok else "DIFFERS"))
        out.write(r.stderr.decode())
        out.flush()
        print("run", k, round(dt, 3), ok)
