#!/usr/bin/env python3
"""Two 512-pair shards of the north-star batch in flight on one MI355X (two host threads, a context and a stream each) against
the same shards one after the other: does a stream of small batches fill the device better when consecutive shards overlap?
(the shard one of eight ranks aligns under --scaling strong runs its four passes one after the other on a
device none of them fills.)  Resident batches (no staging); prints ms per shard for each setting."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exonerate_amd as ex
from exonerate_amd import workloads

N, REPS = 512, int(os.environ.get("PROBE_REPS", "6"))
model = ex.Model("est2genome")
shards = [workloads.est2genome_pairs(N, 1000, 100000, first=k * N) for k in range(2)]


def setup():
    engs = [ex.Engine(0) for _ in range(2)]
    for e in engs:
        e.own_stream()
    bs = [ex.ResidentBatch(e, model, s) for e, s in zip(engs, shards)]
    for b in bs:
        b.run(2); b.run(2)
    return engs, bs


def serial(bs):
    t0 = time.perf_counter()
    for _ in range(REPS):
        for b in bs:
            b.run(2)
    return (time.perf_counter() - t0) / (2 * REPS) * 1e3


def overlapped(bs, lag):
    def loop(b, delay):
        time.sleep(delay)
        for _ in range(REPS):
            b.run(2)
    th = [threading.Thread(target=loop, args=(b, k * lag)) for k, b in enumerate(bs)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return (time.perf_counter() - t0) / (2 * REPS) * 1e3


engs, bs = setup()
s = serial(bs)
print("%-40s serial %.1f ms/shard   two in flight %.1f (lag 0) %.1f (lag 40 ms)" % (
    " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("C4GPU_")) or "defaults",
    s, overlapped(bs, 0.0), overlapped(bs, 0.04)), flush=True)
a0 = bs[0].alignment(0); a1 = bs[1].alignment(0)
print("   sample:", a0.score if a0 else None, a1.score if a1 else None)
