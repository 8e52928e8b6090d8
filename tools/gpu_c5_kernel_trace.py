"""rocprofv3 kernel trace of config 5's heuristic leg through the drop-in (which kernels ran when, on which queue): written
to gpurun_out/c5_ktrace/.  Usage (GPU box): cd /tmp && TMPDIR=/tmp python $GRAFT_REPO_ROOT/tools/gpu_c5_kernel_trace.py"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exonerate_amd import workloads

exe = os.path.join(ROOT, "integration", "_build", "exonerate-gpu")
want = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_c5_heuristic.json")))
out = os.path.join(ROOT, "gpurun_out", "c5_ktrace")
with tempfile.TemporaryDirectory() as d:
    qf, tf = workloads.write_c5_heuristic_input(d)
    env = dict(os.environ, C4GPU_VERBOSE="1", C4GPU_TRACE="1")
    subprocess.run([exe] + want["args"] + [qf, tf], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)      # warm the page cache
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "--", exe] + want["args"] + [qf, tf],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
    open(os.path.join(ROOT, "gpurun_out", "c5_ktrace.err"), "w").write(r.stderr.decode())
    print("rc", r.returncode)
