#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the tracked files
profiles/<tag>_kernel_stats.csv, <tag>_pmc.csv, <tag>_bench.json and profiles/traffic_latest.json."""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r01_c"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
LAUNCHES_IN_PMC_RUN = 2          # bench.py --steps 1 --warmup 1

shutil.copy(glob.glob(src + "/trace/*/*_kernel_stats.csv")[0], f"{dst}/{tag}_kernel_stats.csv")
shutil.copy(src + "/bench.json", f"{dst}/{tag}_bench.json")

def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]

agg = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for f in glob.glob(src + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[k]["dispatches:" + r["Counter_Name"]] += 1
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                   r["Workgroup_Size"], r["Grid_Size"])
names = ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
         "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS"]
with open(f"{dst}/{tag}_pmc.csv", "w") as o:
    o.write("# rocprofv3 --pmc, three separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*), python bench.py --steps 1 --warmup 1;\n"
            "# values are SUMS over the dispatches of that kernel in the run (column 'dispatches'); FETCH/WRITE in KiB, SQ_* in quad-cycles\n")
    o.write("kernel,dispatches,vgpr,agpr,sgpr,lds_bytes,scratch_bytes,workgroup,grid," + ",".join(names) + "\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        o.write('"%s",%d,%s,' % (k, int(v.get("dispatches:SQ_WAVES", v.get("dispatches:FETCH_SIZE", 0))), ",".join(meta[k]))
                + ",".join("%.0f" % v.get(n, 0) for n in names) + "\n")

bench = json.loads(open(src + "/bench.json").read().strip().splitlines()[-1])
cfg = bench["config"]
reg = next(k for k in agg if ", 2, true, true" in k)
v = agg[reg]
n = LAUNCHES_IN_PMC_RUN
stats = {r["Name"]: r for r in csv.DictReader(open(f"{dst}/{tag}_kernel_stats.csv"))}
avg_ns = next(float(r["AverageNs"]) for name, r in stats.items() if short(name) == reg)
waves = v["SQ_WAVES"] / n
out = {
    "source": f"profiles/{tag}_pmc.csv, profiles/{tag}_kernel_stats.csv (tools/profile_round.sh, tools/summarise_profile.py)",
    "config": {"pairs_per_gpu": cfg["pairs_per_gpu"], "query_len": cfg["query_len"], "target_len": cfg["target_len"]},
    "kernel": reg,
    "fetch_kib": v["FETCH_SIZE"] / n, "write_kib": v["WRITE_SIZE"] / n,
    "fetch_correction": 2.0,
    "bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / n,
    "rocprof_avg_launch_ms": avg_ns / 1e6,
    "valu": {
        "insts_per_launch": v["SQ_INSTS_VALU"] / n,
        "lane_ops_per_cell": v["SQ_INSTS_VALU"] / n * 64 / (cfg["pairs_per_gpu"] * (cfg["query_len"] + 1) * (cfg["target_len"] + 1)),
        "active_frac_of_wave_cycles": v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"],
        "wait_frac_of_wave_cycles": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
        "waves_per_launch": waves,
        "simd_issue_utilisation": (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]) * waves / 1024.0,
        # VALU issue roofline: a wave64 VALU instruction occupies its SIMD for 4 cycles; 1024 SIMDs at 2.4 GHz
        "issue_roofline_frac": (v["SQ_INSTS_VALU"] / n) * 4.0 / (1024.0 * (avg_ns * 1e-9) * 2.4e9),
    },
    "note": "fetch_kib / write_kib are the raw counters; bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE: the gfx950 "
            "FETCH_SIZE under-count of MI355X_MICROARCH.md (HBM section), calibrated on this access pattern as the guide "
            "asks: every input byte (Q + T + 16 T per pair = 6.97 GB per launch) has to be fetched at least once, and the raw "
            "counter reads 3.5 GB, i.e. one half. SQ_* are per-wave quad-cycles; "
            "simd_issue_utilisation = VALU-active share of a wave's cycles x resident waves per SIMD (1024 SIMDs).",
}
json.dump(out, open(f"{dst}/traffic_latest.json", "w"), indent=1)
print(json.dumps(out, indent=1))
