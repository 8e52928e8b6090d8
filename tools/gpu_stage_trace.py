"""C4GPU_TRACE marks of one resident step and one streaming step (next batch staged behind) of the north-star batch."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exonerate_amd as ex
from exonerate_amd import workloads
from concurrent.futures import ProcessPoolExecutor
n = 4096
def gen(a):
    return workloads.est2genome_pairs(a[1], 1000, 100000, first=a[0])
with ProcessPoolExecutor(max_workers=32) as pool:
    chunks = list(pool.map(gen, [(b * n + c, 64) for b in range(2) for c in range(0, n, 64)]))
batches = [[p for ch in chunks[b * 64:(b + 1) * 64] for p in ch] for b in range(2)]
eng = ex.Engine(0); model = ex.Model("est2genome")
stage = ex.Stage(eng, model)
stage.load(batches[0]); batch = ex.ResidentBatch(eng, model, batches[0][:2]); batch.swap(stage)
batch.run(2); stage.load(batches[1]); batch.swap(stage); batch.run(2); stage.load(batches[0]); batch.swap(stage); batch.run(2)
os.environ["C4GPU_TRACE"] = "1"
print("=== resident step", file=sys.stderr, flush=True)
t0 = time.perf_counter(); batch.run(2); print("=== %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
stage.load(batches[1])
for k in range(2):
    print("=== streaming step", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    batch.swap(stage)
    th = threading.Thread(target=lambda: stage.load(batches[k % 2])); th.start()
    batch.run(2)
    print("=== %.1f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
    th.join()
