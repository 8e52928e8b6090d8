"""Where the staging time of a north-star batch goes (bench.py's staging_ms: ResidentBatch = flattening in Python +
c4gpu_batch_create: upload, residue coding, splice arrays): six creations in one process, C4GPU_TRACE marks of the library
on stderr.  usage (GPU box): python tools/gpu_staging_probe.py > gpurun_out/staging_probe.log 2>&1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["C4GPU_TRACE"] = "1"
import torch
import exonerate_amd as ex
from exonerate_amd import workloads

pairs = workloads.est2genome_pairs(4096)
eng = ex.Engine(0)
model = ex.Model("est2genome")
for rep in range(6):
    t0 = time.perf_counter()
    arr, keep = ex._pairs(pairs)
    t1 = time.perf_counter()
    h = ex._lib().c4gpu_batch_create(eng.ctx, model.c, model.params, arr, len(pairs))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("creation %d: python flattening %.1f ms, c4gpu_batch_create %.1f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    sys.stderr.flush()
    ex._lib().c4gpu_batch_destroy(h)
eng.close()
