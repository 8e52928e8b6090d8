import os, sys, json, time
sys.path.insert(0, os.getcwd())
import exonerate_amd as ex
import bench
eng = ex.Engine(0)
out = bench.other_configs(ex, eng)
for k, v in out.items():
    print(k, "%.4e cells/s" % v["value"], "%.1f ms" % v["ms_per_pass"], v["kernel_ms"], v["checked"][:40])
