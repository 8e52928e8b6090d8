"""BASELINE configurations 2, 3 and 5 (exhaustive shape) at their full sizes through bench.other_configs: cells/s, kernel times per
kind, samples against tests/golden/bench_configs.json -- a quick look at the protein and affine kernels without the whole bench line."""
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import exonerate_amd as ex
import bench
eng = ex.Engine(0)
out = bench.other_configs(ex, eng)
for k, v in out.items():
    print(k, "%.4e cells/s" % v["value"], "%.1f ms" % v["ms_per_pass"], v["kernel_ms"], v["checked"][:40])
