#!/usr/bin/env python3
"""Time the drop-in binary in the HEURISTIC mode (BSDP: --gappedextension no, with and without --refine region) against
the unmodified reference on north-star-shaped input: N cDNAs (1 kb) x N genomic windows (100 kb), all against all, both
strands.  Rows: the reference (whole input: the heuristic mode is fast), exonerate-gpu with the BSDP / refinement batches
(integration/c4gpu_bsdp.c), and exonerate-gpu with that seam off (every refinement its own device call).  Outputs are
compared byte for byte.  Writes a markdown table to stdout."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exonerate_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/heur"
os.makedirs(out, exist_ok=True)
pairs = workloads.est2genome_pairs(n, 1000, 100000, seed=20260928 + 4)
for path, recs in ((out + "/q.fa", [("cdna%d" % i, p[0]) for i, p in enumerate(pairs)]),
                   (out + "/t.fa", [("win%d" % i, p[1]) for i, p in enumerate(pairs)])):
    with open(path, "w") as f:
        for name, s in recs:
            f.write(">%s\n%s\n" % (name, s.decode()))
gpu_exe = ROOT + "/integration/_build/exonerate-gpu"
cpu_exe = ROOT + "/oracle/_ref/exonerate-compiled"


def run(exe, extra, env=None):
    e = dict(os.environ, C4GPU_VERBOSE="1")
    e.update(env or {})
    mode = [] if "--gappedextension" in extra else ["--gappedextension", "no"]
    args = ["-m", "est2genome"] + mode + ["--showalignment", "no", "--showvulgar", "yes", "-V", "0"] + extra
    t0 = time.perf_counter()
    r = subprocess.run([exe] + args + [out + "/q.fa", out + "/t.fa"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    return r.stdout.decode(), dt, r.stderr.decode()


run(gpu_exe, ["-S", "no"])                                    # warm-up (HIP module load)
print("# Heuristic (BSDP) mode of the drop-in binary, %d cDNAs x %d windows of 100 kb, all against all, both strands\n" % (n, n))
print("| flags | binary | wall s | alignments | sub-DP calls answered by device batches | refinements from batches |")
print("|---|---|---|---|---|---|")
for extra in (["-S", "no"], [], ["--refine", "region", "-S", "no"], ["--refine", "region"]):
    ref, t_ref, _ = run(cpu_exe, extra)
    gpu, t_gpu, err = run(gpu_exe, extra)
    off, t_off, _ = run(gpu_exe, extra, {"C4GPU_BSDP_OFF": "1"})
    hoff, t_hoff, _ = run(gpu_exe, extra, {"C4GPU_HSP_OFF": "1"})
    assert gpu == ref and off == ref and hoff == ref, "outputs differ for %r" % (extra,)
    hs = re.search(r"c4gpu hsp: (\d+) word hits of \d+ HSP sets extended in (\d+) device batch", err)
    m = re.search(r"(\d+) of (\d+) score calls and (\d+) of (\d+) path calls served from them; (\d+) of (\d+) refinements", err)
    served = "%d of %d" % (int(m.group(1)) + int(m.group(3)), int(m.group(2)) + int(m.group(4))) if m else "?"
    refined = "%s of %s" % (m.group(5), m.group(6)) if m else "?"
    flags = " ".join(extra) or "(default)"
    nal = ref.count("vulgar:")
    print("| `%s` | reference (compiled Viterbi, 1 core) | %.2f | %d | - | - |" % (flags, t_ref, nal))
    print("| `%s` | exonerate-gpu, BSDP + refinement batches | %.2f | %d | %s | %s |" % (flags, t_gpu, nal, served, refined))
    print("| `%s` | exonerate-gpu, seam off (one device call per refinement) | %.2f | %d | - | - |" % (flags, t_off, nal))
    print("| `%s` | exonerate-gpu, seeding on the host (C4GPU_HSP_OFF=1), rest as row 2 | %.2f | %d | - | - |" % (flags, t_hoff, nal))
    if hs:
        print("| `%s` | (row 2: %s word hits extended in %s device launches) | | | | |" % (flags, hs.group(1), hs.group(2)))
print("\nAll outputs byte-identical to the reference's.")

# the default mode: --gappedextension yes = SDP (boundary flavour for est2genome), integration/c4gpu_sdp.c
print("\n# Default heuristic mode (--gappedextension yes: seeding + SDP), same input\n")
print("| flags | binary | wall s | alignments | pairs served from SDP device batches |")
print("|---|---|---|---|---|")
for extra in (["--gappedextension", "yes", "-S", "no"], ["--gappedextension", "yes"]):
    ref, t_ref, _ = run(cpu_exe, extra)
    times = []
    for rep in range(3):
        gpu, t_gpu, err = run(gpu_exe, extra)
        assert gpu == ref, "outputs differ for %r" % (extra,)
        times.append(t_gpu)
    m = re.search(r"c4gpu sdp: (\d+) pairs in (\d+) flush\(es\): (\d+) served from device batches \((\d+) alignments\); batches (\d+) ms", err)
    flags = " ".join(extra)
    nal = ref.count("vulgar:")
    print("| `%s` | reference (compiled scheduler, 1 core) | %.2f | %d | - |" % (flags, t_ref, nal))
    print("| `%s` | exonerate-gpu (every pair through the sparse SDP wavefront kernels, no size limit) | %.2f (median of 3) | %d | %s |" % (flags, sorted(times)[1], nal,
          ("%s of %s in %s flushes, %s ms in the batches" % (m.group(3), m.group(1), m.group(2), m.group(5))) if m else "?"))
print("\nAll outputs byte-identical to the reference's.")
