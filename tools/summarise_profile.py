#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the tracked files
profiles/<tag>_kernel_stats.csv, <tag>_pmc.csv, <tag>_bench.json and profiles/traffic_latest.json."""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r01_c"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
LAUNCHES_IN_PMC_RUN = 2          # STEPS of the PMC run: bench.py --steps 1 --warmup 1 (also the fallback dispatch count)

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
            "# values are SUMS over the dispatches of that kernel in the run (column 'dispatches'); FETCH/WRITE in KiB, SQ_* in quad-cycles\n"
            "# vgpr: rocprofv3's VGPR_Count column, which for these wave64 kernels is HALF the per-lane register count of the code object\n"
            "# (.vgpr_count in the kernel's metadata, llvm-readelf --notes: 84 here <-> 168 for the packed score pass): waves per SIMD =\n"
            "# floor(512 / (2 x vgpr)); lds_bytes and scratch_bytes are per workgroup / per lane as rocprofv3 reports them\n")
    o.write("kernel,dispatches,vgpr,agpr,sgpr,lds_bytes,scratch_bytes,workgroup,grid," + ",".join(names) + "\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        o.write('"%s",%d,%s,' % (k, int(v.get("dispatches:SQ_WAVES", v.get("dispatches:FETCH_SIZE", 0))), ",".join(meta[k]))
                + ",".join("%.0f" % v.get(n, 0) for n in names) + "\n")

bench = json.loads(open(src + "/bench.json").read().strip().splitlines()[-1])
cfg = bench["config"]
stats = {r["Name"]: r for r in csv.DictReader(open(f"{dst}/{tag}_kernel_stats.csv"))}
# the dominant kernel of the step: largest total time among the Viterbi kernels
dom = max((r for name, r in stats.items() if "viterbi_kernel" in name or "viterbi16_kernel" in name), key=lambda r: float(r["TotalDurationNs"]))
reg = short(dom["Name"])
v = agg[reg]
n = int(v.get("dispatches:SQ_WAVES", v.get("dispatches:FETCH_SIZE", LAUNCHES_IN_PMC_RUN)))
avg_ns = float(dom["AverageNs"])
waves = v["SQ_WAVES"] / n
# template arguments: viterbi_kernel_mw<Model, R, MODE, ...>
mode = 0 if "viterbi16" in reg else int(reg.split("<")[1].split(",")[2])      # the packed 16-bit kernel is a FIND_SCORE pass
vj = json.load(open(f"{dst}/valu_issue_latest.json"))
wave_cycles = 4.0 * v["SQ_WAVE_CYCLES"] / max(v["SQ_WAVES"], 1)          # shader cycles one wave lives (quad-cycle counter)
waves_per_simd = waves / vj["simds"]
ipc = (v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1)) / wave_cycles * waves_per_simd
# every pass of the step: wave-instructions per PAIR of the batch (a dispatch covers pairs_per_gpu / launches_per_step pairs),
# registers, waiting share, the trace's launch time -- what bench.py's roofline.kernels sets this run's launch times against
sys.path.insert(0, root)
from exonerate_amd.srchash import csrc_hash
lps = max(1, bench["roofline"].get("launches_per_step", 2))
pairs_per_dispatch = cfg["pairs_per_gpu"] / lps
def is_path(k):
    if not k.startswith("c4k::viterbi_kernel<"): return False
    a = [x.strip() for x in k.split("<", 1)[1].rstrip(">").split(",")]
    return len(a) > 2 and a[2] == "1"
groups = {"score": lambda k: "viterbi16_kernel_mw" in k or ("viterbi_kernel_mw" in k), "windows": lambda k: "win16_kernel" in k,
          "checkpoint": lambda k: "ckpt16_kernel" in k, "path": is_path}
kernels = {}
for gname, pred in groups.items():
    ks = [k for k in agg if pred(k)]
    if not ks: continue
    k = max(ks, key=lambda k: agg[k].get("SQ_WAVE_CYCLES", 0))
    a = agg[k]
    nd = int(a.get("dispatches:SQ_WAVES", 0)) or 1
    tr = [r for name, r in stats.items() if short(name) == k]
    vg = int(meta[k][0]) * 2
    kernels[gname] = {"kernel": k, "dispatches_in_pmc_run": nd,
                      "wave_insts_per_pair": a["SQ_INSTS_VALU"] / nd / pairs_per_dispatch,
                      "wait_frac": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES") else None,
                      "vgprs_per_lane": vg, "waves_per_simd": 512 // vg if vg else None,
                      "lds_bytes": int(meta[k][3]), "scratch_bytes_per_lane": int(meta[k][4]),
                      "rocprof_avg_launch_ms": float(tr[0]["AverageNs"]) / 1e6 if tr else None,
                      "valu_frac_of_nominal_in_trace": (a["SQ_INSTS_VALU"] / nd) / (float(tr[0]["AverageNs"]) * 1e-9) / (0.5 * 1024 * 2.4e9) if tr else None}
out = {
    "source": f"profiles/{tag}_pmc.csv, profiles/{tag}_kernel_stats.csv (tools/profile_round.sh, tools/summarise_profile.py)",
    "csrc_hash": open(src + "/csrc_hash.txt").read().strip() if os.path.exists(src + "/csrc_hash.txt") else csrc_hash(), "kernels": kernels,
    "config": {"pairs_per_gpu": cfg["pairs_per_gpu"], "query_len": cfg["query_len"], "target_len": cfg["target_len"]},
    "kernel": reg, "mode": mode, "dispatches_in_pmc_run": n,
    "fetch_kib": v["FETCH_SIZE"] / n, "write_kib": v["WRITE_SIZE"] / n,
    "fetch_correction": 2.0,
    "bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / n,
    "rocprof_avg_launch_ms": avg_ns / 1e6,
    "valu": {
        "insts_per_launch": v["SQ_INSTS_VALU"] / n,
        # cells per launch: the PMC run makes LAUNCHES_IN_PMC_RUN steps (warm-up + one); a large batch runs as two halves on
        # two launch lanes, i.e. two dispatches per step of half the pairs each
        # (every dispatch of the run covers pairs_per_gpu / launches_per_step pairs: round 5's bench stages and aligns more batches
        # per run than the two timed-plus-warm-up steps -- resident passes, the staging measurement -- all of the full size)
        "lane_ops_per_cell": v["SQ_INSTS_VALU"] / n * 64 / (cfg["pairs_per_gpu"] / max(1, bench["roofline"].get("launches_per_step", 2))
                                                            * (cfg["query_len"] + 1) * (cfg["target_len"] + 1)),
        "active_frac_of_wave_cycles": v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"],
        "wait_frac_of_wave_cycles": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
        "waves_per_launch": waves,
        "simd_issue_utilisation": (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]) * waves / 1024.0,
        # measured against the issue rates of profiles/valu_issue_latest.json (tools/valu_issue_microbench.hip):
        # wave-instructions per shader cycle per SIMD from the counters alone (SQ_WAVE_CYCLES gives the cycles)
        "waves_per_simd": waves_per_simd,
        "cycles_per_inst_per_wave": wave_cycles / (v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1)),
        "wave_inst_per_clk_per_simd": ipc,
        "effective_clock_ghz": wave_cycles / (avg_ns * 1e-9) / 1e9,
        "frac_of_mix_rate_at_this_occupancy": ipc / vj["wave_inst_per_clk_per_simd_by_waves"].get(str(int(round(waves_per_simd))), vj["peak_wave_inst_per_clk_per_simd"]),
        "frac_of_mix_peak_8_waves": ipc / vj["peak_wave_inst_per_clk_per_simd"],
    },
    "note": "fetch_kib / write_kib are the raw counters; bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE: FETCH_SIZE reads "
            "exactly one half of the bytes fetched at 1, 2, 4 and 16 B per lane, streaming and sliding alike, and WRITE_SIZE "
            "is exact (profiles/r02_fetch_calibration.md: known-size reads in this kernel's own access widths). SQ_* are "
            "per-wave quad-cycles; simd_issue_utilisation = VALU-active share of a wave's cycles x resident waves per SIMD.",
}
# bench.py's roofline block reads traffic_latest.json for the launches IT times (two launch lanes): a one-lane profile
# (C4GPU_LANES=1, tag *_lanes1) has launches twice the size and is kept under its own name only
json.dump(out, open(f"{dst}/traffic_latest.json" if "lanes1" not in tag else f"{dst}/{tag}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
