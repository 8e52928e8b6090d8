#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_sq/ (tools/profile_sq_breakdown.sh) -> profiles/<tag>_sq.csv: per kernel the sums of every SQ
counter that was collected, and per-wave-cycle / per-VALU-instruction ratios."""
import csv, glob, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r03_f"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_%s_sq" % tag)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(src + "/pass*/*/*_counter_collection.csv"):
    p = f.split("/pass")[1].split("/")[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"] + ("" if r["Counter_Name"] not in ("SQ_WAVES", "SQ_WAVE_CYCLES") else "@" + p)] += float(r["Counter_Value"])
names = sorted({n for v in agg.values() for n in v})
with open(os.path.join(root, "profiles", tag + "_sq.csv"), "w") as o:
    o.write("# rocprofv3 --pmc, one pass per counter group (tools/profile_sq_breakdown.sh), python bench.py --steps 1 --warmup 1; sums over\n"
            "# the dispatches of a kernel in the run; SQ_WAVES / SQ_WAVE_CYCLES are repeated in every pass (@pass)\n")
    o.write("kernel," + ",".join(names) + "\n")
    for k, v in sorted(agg.items(), key=lambda kv: -max(kv[1].get("SQ_WAVE_CYCLES@1", 0), 0)):
        if not any(t in k for t in ("viterbi", "win16", "ckpt16")):
            continue
        o.write('"%s",' % k + ",".join("%.0f" % v.get(n, 0) for n in names) + "\n")
print(open(os.path.join(root, "profiles", tag + "_sq.csv")).read()[:3000])
