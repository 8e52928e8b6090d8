#!/usr/bin/env python3
"""Experiment: the north-star batch as ONE resident batch on one stream against the same pairs as TWO (or four) half batches on
their own streams, driven from host threads (the persistent kernels of one half fill the tails of the other's).
Prints ms per step (all 4 096 pairs aligned per step in every form)."""
import sys, time, os, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import exonerate_amd as ex
from exonerate_amd import workloads

n = int(os.environ.get("PAIRS", "4096"))
pairs = workloads.est2genome_pairs(n, 1000, 100000)
model = ex.Model("est2genome")
steps = 3
for parts in [int(x) for x in os.environ.get("PARTS", "1,2,4,1,2").split(",")]:
    streams = [torch.cuda.Stream() for _ in range(parts)]
    engines = [ex.Engine(0, stream=s.cuda_stream) for s in streams]
    per = n // parts
    batches = [ex.ResidentBatch(engines[k], model, pairs[k * per:(k + 1) * per]) for k in range(parts)]

    def step():
        th = [threading.Thread(target=b.run, args=(2, 32)) for b in batches[1:]]
        for t in th:
            t.start()
        batches[0].run(2, 32)
        for t in th:
            t.join()

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ok = sum(1 for b in batches for i in range(0, per, max(1, per // 8)) if b.alignment(i) is not None)
    print("parts %d: %.1f ms per step (%d sampled alignments present)" % (parts, dt * 1e3, ok), flush=True)
    for b in batches:
        b.close()
    for e in engines:
        e.close()
