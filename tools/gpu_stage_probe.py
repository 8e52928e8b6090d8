"""What staging a north-star batch costs through a c4gpu_stage (page-locked gather, sliced upload, coding, splice arrays,
packed splice array; every buffer reused), alone and while another batch runs; and what a step costs with the next batch
staged behind it.  usage (GPU box): python tools/gpu_stage_probe.py [pairs] > gpurun_out/stage_probe.log 2>&1"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import exonerate_amd as ex
from exonerate_amd import workloads
from concurrent.futures import ProcessPoolExecutor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NB = 4


def gen(args):
    first, cnt = args
    return workloads.est2genome_pairs(cnt, 1000, 100000, first=first)


t0 = time.perf_counter()
with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex_:
    chunks = list(ex_.map(gen, [(b * n + c, min(64, n - c)) for b in range(NB) for c in range(0, n, 64)]))
per = (n + 63) // 64
batches = [[p for ch in chunks[b * per:(b + 1) * per] for p in ch] for b in range(NB)]
print("generated %d batches of %d pairs in %.1f s" % (NB, n, time.perf_counter() - t0), flush=True)

eng = ex.Engine(0)
model = ex.Model("est2genome")
stage = ex.Stage(eng, model)
if os.environ.get("PROBE_TRACE"):
    os.environ["C4GPU_TRACE"] = "1"
for rep in range(5):
    t0 = time.perf_counter()
    ms = stage.load(batches[rep % NB])
    print("load %d alone: %.1f ms wall (library %.1f ms)" % (rep, (time.perf_counter() - t0) * 1e3, ms), flush=True)
    if rep == 0:
        batch = ex.ResidentBatch(eng, model, batches[0][:2])
    batch.swap(stage)                       # so that the next load reuses the previous buffers, as in a stream of batches
os.environ.pop("C4GPU_TRACE", None)
names = {0: "score", 1: "path", 2: "region", 3: "checkpoint"}


def kstats():
    return " ".join("%s %.1f" % (names[m], batch.kernel_stats(m, reset=True)["ms"]) for m in range(4))


kstats()
# resident: every batch, the same one again and again
batch.run(2)
resident = {}
for k in range(NB):
    stage.load(batches[k]); batch.swap(stage)
    batch.run(2); batch.export(); kstats()
    ts = []
    for rep in range(2):
        t0 = time.perf_counter()
        batch.run(2); batch.export()
        ts.append((time.perf_counter() - t0) * 1e3)
        print("step, batch %d resident: %.1f ms  [kernel ms: %s]" % (k, ts[-1], kstats()), flush=True)
    resident[k] = min(ts)
# a fresh batch per step, staged one after the other
for rep in range(4):
    t0 = time.perf_counter()
    stage.load(batches[rep % NB]); batch.swap(stage)
    t1 = time.perf_counter()
    batch.run(2); batch.export()
    print("step, batch %d staged in front: %.1f ms (load %.1f)  [kernel ms: %s]" % (rep % NB, (time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3, kstats()), flush=True)
# a fresh batch per step, the next one staged behind the current one
stage.load(batches[0])
tot = base = 0.0
for rep in range(8):
    t0 = time.perf_counter()
    batch.swap(stage)
    res = {}

    def bg(k=rep):
        c0 = time.perf_counter()
        res["lib"] = stage.load(batches[(k + 1) % NB])
        res["wall"] = (time.perf_counter() - c0) * 1e3

    th = threading.Thread(target=bg); th.start()
    batch.run(2); batch.export()
    t1 = time.perf_counter()
    th.join()
    dt = (time.perf_counter() - t0) * 1e3
    if rep >= 4:
        tot += dt; base += resident[rep % NB]
    print("step, batch %d, next batch staged behind: %.1f ms (run+export %.1f, load in the background %.1f wall / %.1f library)  [kernel ms: %s]"
          % (rep % NB, dt, (t1 - t0) * 1e3, res["wall"], res["lib"], kstats()), flush=True)
print("streaming - resident over the last four steps: %.1f ms per step" % ((tot - base) / 4))
batch.close(); stage.close(); eng.close()
