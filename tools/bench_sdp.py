"""SDP on the device (c4gpu_sdp_batch) on BASELINE config 1's shape: N proteins of ~300 aa against one 10 kaa target
that holds diverged copies of them, affine:local — the device call against the oracle's restatement (1 core) and, for
scale, the reference binary's whole run on the same FASTA files.  Prints a markdown table.
usage: python tools/bench_sdp.py [queries=100]"""
import os, random, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import exonerate_amd as ex
import oracle_lib

AA = "ARNDCQEGHILKMFPSTWYV"


def mut(rng, s, rate):
    out = []
    for c in s:
        x = rng.random()
        if x < rate:
            out.append(rng.choice(AA))
        elif x < rate * 1.2:
            continue
        elif x < rate * 1.4:
            out.append(c + rng.choice(AA))
        else:
            out.append(c)
    return "".join(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = random.Random(5)
    queries = ["".join(rng.choice(AA) for _ in range(rng.randint(250, 350))) for _ in range(n)]
    target = ""
    for q in rng.sample(queries, min(n, 24)):                       # 24 diverged copies + filler = ~10 kaa
        target += "".join(rng.choice(AA) for _ in range(rng.randint(40, 160))) + mut(rng, q, 0.15)
    target = target[:10000]
    params = ex.default_params()
    model = ex.Model("affine:local", query_alphabet=ex.ALPHABET_PROTEIN, target_alphabet=ex.ALPHABET_PROTEIN, params=params)
    w = 4
    pairs, hsps = [], []
    for q in queries:
        words = {}
        for i in range(len(q) - w + 1):
            words.setdefault(q[i:i + w], []).append(i)
        seeds = [(i, j) for j in range(len(target) - w + 1) for i in words.get(target[j:j + w], ())]
        h = oracle_lib.hsp_set(params, "protein2protein", q.encode(), target.encode(), w, 20, 30, seeds) if seeds else []
        if h:
            pairs.append((q, target)); hsps.append(h)
    eng = ex.Engine(0)
    eng.sdp(model, pairs[:1], hsps[:1], 1, 1, 50, 100, 4)                                       # context, module load
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        got = eng.sdp(model, pairs, hsps, 1, 1, 50, 100, 4)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    t0 = time.perf_counter()
    exp = [oracle_lib.sdp(model.c, model.params, q.encode(), t.encode(), h, 1, 1, 50, True, 100, 4)[1] for (q, t), h in zip(pairs, hsps)]
    cpu = time.perf_counter() - t0
    same = all([a.as_dict() for a in g] == e for g, e in zip(got, exp))
    cells = sum((len(q) + 1) * (len(t) + 1) for q, t in pairs)
    print("| what | pairs with HSPs | HSPs | alignments | wall ms | lattice cells/s (2 passes) |")
    print("|---|---|---|---|---|---|")
    print(f"| c4gpu_sdp_batch: {n} proteins of ~300 aa x one 10 kaa target (host buffers in, alignments out) | {len(pairs)} | "
          f"{sum(len(h) for h in hsps)} | {sum(len(g) for g in got)} | {best * 1e3:.1f} | {2 * cells / best:.3g} |")
    print(f"| oracle_sdp (scalar C restatement of the reference's sparse scheduler, 1 core) | {len(pairs)} | - | "
          f"{sum(len(e) for e in exp)} | {cpu * 1e3:.1f} | - |")
    ref = os.path.join(ROOT, "oracle", "_ref", "exonerate-compiled")
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "q.fa"), "w") as f:
                for k, q in enumerate(queries):
                    f.write(">q%d\n%s\n" % (k, q))
            with open(os.path.join(d, "t.fa"), "w") as f:
                f.write(">t\n%s\n" % target)
            t0 = time.perf_counter()
            r = subprocess.run([ref, "-m", "affine:local", "--showalignment", "no", "--showvulgar", "yes", "-V", "0",
                                os.path.join(d, "q.fa"), os.path.join(d, "t.fa")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            print(f"| reference binary, whole run on the same FASTA files (its own seeding + SDP, 1 core) | {n} queries | - | "
                  f"{r.stdout.decode().count('vulgar:')} | {dt * 1e3:.1f} | - |")
            gpu = os.path.join(ROOT, "integration", "_build", "exonerate-gpu")
            if os.path.exists(gpu):
                args = ["-m", "affine:local", "--showalignment", "no", "--showvulgar", "yes", "-V", "0",
                        os.path.join(d, "q.fa"), os.path.join(d, "t.fa")]
                subprocess.run([gpu] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)           # module load
                t0 = time.perf_counter()
                g = subprocess.run([gpu] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, C4GPU_VERBOSE="1"))
                dt = time.perf_counter() - t0
                note = [l for l in g.stderr.decode().splitlines() if "c4gpu sdp:" in l or "c4gpu hsp:" in l]
                print(f"| exonerate-gpu, whole run (seeding and SDP seams; output {'identical' if g.stdout == r.stdout else 'DIFFERENT'}) | "
                      f"{n} queries | - | {g.stdout.decode().count('vulgar:')} | {dt * 1e3:.1f} | - |")
                for l in note:
                    print("|   " + l.split("Message:")[-1].strip().replace("|", "/") + " | | | | | |")
    print(f"\ndevice alignments identical to the oracle's: {'yes' if same else 'NO'}")
    eng.close()


if __name__ == "__main__":
    main()
