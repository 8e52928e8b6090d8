#!/usr/bin/env python3
"""Where the drop-in's start-up goes: dlopen of libc4gpu.so, c4gpu_ctx_create, the first launches; and the wall time of
exonerate-gpu against the reference on small heuristic inputs (config 1's shape; the 32 x 32 est2genome heuristic run)."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
t0 = time.perf_counter()
lib = C.CDLL(os.path.join(ROOT, "exonerate_amd", "libc4gpu.so"))
t1 = time.perf_counter()
lib.c4gpu_ctx_create.restype = C.c_void_p
ctx = lib.c4gpu_ctx_create(0)
t2 = time.perf_counter()
print("dlopen libc4gpu.so %.1f ms; c4gpu_ctx_create %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
import exonerate_amd as ex
eng = ex.Engine(0)
model = ex.Model("est2genome")
for k in range(3):
    a = time.perf_counter()
    eng.find_score(model, [("ACGTACGTAGCTAGCTAGCTAGCATCGATCG", "ACGTACGTAGCTAGCTAGGGCTAGCATCGATCG")])
    print("find_score call %d: %.1f ms" % (k, (time.perf_counter() - a) * 1e3))
eng.close()
import test_integration_bsdp_host as hb
import tempfile, pathlib
gpu_exe, cpu_exe = hb.GPU_EXE, hb.CPU_EXE
def wall(exe, args, env=None, reps=3):
    best = 1e9
    for _ in range(reps):
        a = time.perf_counter()
        r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
        best = min(best, time.perf_counter() - a)
        assert r.returncode == 0, r.stderr.decode()[-500:]
    return best, r.stdout
with tempfile.TemporaryDirectory() as d:
    from exonerate_amd import workloads
    import random
    rng = random.Random(5)
    AA = "ARNDCQEGHILKMFPSTWYV"
    qs = ["".join(rng.choice(AA) for _ in range(rng.randint(250, 350))) for _ in range(100)]
    t = ""
    for q in rng.sample(qs, 24):
        t += "".join(rng.choice(AA) for _ in range(rng.randint(40, 160))) + "".join((rng.choice(AA) if rng.random() < 0.15 else c) for c in q)
    open(d + "/q.fa", "w").write("".join(">q%d\n%s\n" % (i, q) for i, q in enumerate(qs)))
    open(d + "/t.fa", "w").write(">t\n%s\n" % t[:10000])
    args = ["-m", "affine:local", "--showalignment", "no", "--showvulgar", "yes", "-V", "0", d + "/q.fa", d + "/t.fa"]
    r, o1 = wall(cpu_exe, args); g, o2 = wall(gpu_exe, args)
    assert o1 == o2
    print("C1 (100 proteins x 10 kaa, affine:local heuristic): reference %.3f s, exonerate-gpu %.3f s" % (r, g))
    gn, _ = wall(gpu_exe, args, {"C4GPU_NO_WARM": "1"})
    gd, _ = wall(gpu_exe, args, {"C4GPU_DISABLE": "1"})
    print("   ... without the background code-object loads %.3f s; with the device switched off (C4GPU_DISABLE) %.3f s" % (gn, gd))
    v = subprocess.run([gpu_exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, C4GPU_VERBOSE="1", C4GPU_TRACE="1"))
    print("   " + "\n   ".join(l for l in v.stderr.decode().splitlines() if "c4gpu" in l and "staging" not in l)[:5000])
    v = subprocess.run([gpu_exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, C4GPU_TRACE="1", C4GPU_DISABLE="1"))
    print("   device off:\n   " + "\n   ".join(l for l in v.stderr.decode().splitlines() if "c4gpu mark" in l)[:3000])
    pairs = workloads.est2genome_pairs(32, 1000, 100000, seed=20260928 + 4)
    open(d + "/q2.fa", "w").write("".join(">c%d\n%s\n" % (i, p[0].decode()) for i, p in enumerate(pairs)))
    open(d + "/t2.fa", "w").write("".join(">w%d\n%s\n" % (i, p[1].decode()) for i, p in enumerate(pairs)))
    for extra in (["--gappedextension", "no", "-S", "no"], ["--gappedextension", "no"], []):
        args = ["-m", "est2genome"] + extra + ["--showalignment", "no", "--showvulgar", "yes", "-V", "0", d + "/q2.fa", d + "/t2.fa"]
        r, o1 = wall(cpu_exe, args); g, o2 = wall(gpu_exe, args)
        assert o1 == o2
        g2, _ = wall(gpu_exe, args, {"C4GPU_SEED_OFF": "1"})
        print("est2genome heuristic 32 x 32 %s: reference %.3f s, exonerate-gpu %.3f s (word scan on the host: %.3f s)" % (" ".join(extra), r, g, g2))
