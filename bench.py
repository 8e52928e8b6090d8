#!/usr/bin/env python3
"""bench.py — the north-star hot path on N MI355X GPUs of one node.

One "step" = one full pass of the hot path (Optimal_find_path: region pass over the whole rectangle,
checkpoint pass and sub-alignment passes over the aligned region, traceback, run-length op lists) over
one resident batch of est2genome pairs (1 kb cDNA x 100 kb genomic).  Pairs are independent, so ranks get
disjoint shards (weak scaling: the per-GPU batch is fixed) and there is no data-path collective; RCCL is
used only for the timing barrier / max.

Prints ONE JSON line on rank 0:
  value = first-pass lattice cells of all ranks / max-over-ranks wall time of the K timed steps
  roofline   = region-pass kernel (the dominant kernel) against the HBM roof (MI355X_MICROARCH.md)
  cpu_baseline = the oracle (oracle/c4_oracle.c, "port") on one pair of the same workload, 1 core
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(args, rank, model, pairs, batch, eng):
    """The CPU path timed on this host, 1 core, on a bounded sample of the same workload.

    Preferred: the REFERENCE ITSELF with its own compiled (code-generated) Viterbi, oracle/_ref/exonerate-compiled
    (built in the build container by `make -C oracle compiled`; travels with the repo snapshot), on pair 0 —
    which also checks the GPU's vulgar line against the reference's at full size.  Otherwise the oracle
    (oracle/c4_oracle.c, an interpreted-style restatement, ~2x slower than the reference's compiled path)."""
    import subprocess, tempfile
    import exonerate_amd as ex
    from exonerate_amd import workloads
    q, t = pairs[0]
    cells = (len(q) + 1) * (len(t) + 1)
    exe = os.path.join(ROOT, "oracle", "_ref", "exonerate-compiled")
    if os.path.exists(exe) and cells <= 3e8:
        try:
            with tempfile.TemporaryDirectory() as d:
                open(os.path.join(d, "q.fa"), "w").write(">qy\n%s\n" % q.decode())
                open(os.path.join(d, "t.fa"), "w").write(">tg\n%s\n" % t.decode())
                c0 = time.perf_counter()
                r = subprocess.run([exe, "-m", "est2genome", "-E", "yes", "-S", "no", "--revcomp", "no",
                                    "--showalignment", "no", "--showvulgar", "yes", "-V", "0",
                                    os.path.join(d, "q.fa"), os.path.join(d, "t.fa")],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                cpu_s = time.perf_counter() - c0
            ref_lines = [l for l in r.stdout.decode().splitlines() if l.startswith("vulgar:")]
            got = batch.alignment(0)
            if r.returncode == 0 and ref_lines:
                same = got is not None and got.vulgar("qy", "tg") == ref_lines[0].strip()
                assert same, "GPU vulgar differs from the reference: %r vs %r" % (got and got.vulgar(), ref_lines[0])
                return {"value": cells / cpu_s, "unit": "cells/s", "cores": 1, "kind": "reference",
                        "sample": "pair 0 of the batch (%d x %d) through the reference's own exonerate (compiled "
                                  "Viterbi, -m est2genome -E yes -S no --revcomp no), %.1f s wall incl. all passes"
                                  % (len(q), len(t), cpu_s),
                        "vulgar_identical_to_gpu": True}
        except (OSError, subprocess.SubprocessError):
            pass
    import oracle_lib
    q, t = workloads.est2genome_pairs(1, args.qlen, min(args.tlen, 20000), first=rank * args.pairs)[0]
    c0 = time.perf_counter()
    exp = oracle_lib.find_path(model.c, model.params, q, t)
    cpu_s = time.perf_counter() - c0
    got = eng.find_path(model, [(q, t)])[0]
    assert got is not None and got.as_dict() == exp, "GPU alignment differs from the oracle"
    return {"value": (len(q) + 1) * (len(t) + 1) / cpu_s, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "1 pair from the same generator (%d x %d), oracle/c4_oracle.c -O3, %.1f s, all passes of "
                      "Optimal_find_path" % (len(q), len(t), cpu_s),
            "checked_bit_exact_vs_gpu": True}


def cpu_baseline_all_cores(pairs, batch):
    """the reference's only multi-core story is N independent processes
    (--querychunkid/--querychunktotal, exonerate.c:64-75): one reference process per host core, each on its own
    pair of the batch, all at once.  Every vulgar line is also compared with the GPU's."""
    import subprocess, tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "exonerate-compiled")
    if not os.path.exists(exe):
        return None
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    n = max(1, min(cores, len(pairs), 64))
    cells = sum((len(pairs[k][0]) + 1) * (len(pairs[k][1]) + 1) for k in range(n))
    with tempfile.TemporaryDirectory() as d:
        for k in range(n):
            open(os.path.join(d, "q%d.fa" % k), "w").write(">qy\n%s\n" % pairs[k][0].decode())
            open(os.path.join(d, "t%d.fa" % k), "w").write(">tg\n%s\n" % pairs[k][1].decode())
        c0 = time.perf_counter()
        procs = [subprocess.Popen([exe, "-m", "est2genome", "-E", "yes", "-S", "no", "--revcomp", "no",
                                   "--showalignment", "no", "--showvulgar", "yes", "-V", "0",
                                   os.path.join(d, "q%d.fa" % k), os.path.join(d, "t%d.fa" % k)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for k in range(n)]
        outs = [p.communicate(timeout=900)[0].decode() for p in procs]
        wall = time.perf_counter() - c0
    same = 0
    for k, o in enumerate(outs):
        ref = [l.strip() for l in o.splitlines() if l.startswith("vulgar:")]
        got = batch.alignment(k)
        assert ref and got is not None and got.vulgar("qy", "tg") == ref[0], "pair %d: GPU vulgar differs from the reference" % k
        same += 1
    return {"value": cells / wall, "unit": "cells/s", "cores": n, "kind": "reference",
            "sample": "pairs 0..%d of the batch, one reference process per core, %.1f s wall" % (n - 1, wall),
            "vulgar_identical_to_gpu": same}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=int(os.environ.get("C4_BENCH_PAIRS", "4096")),
                    help="pairs per GPU (BASELINE: 4096)")
    ap.add_argument("--qlen", type=int, default=1000)
    ap.add_argument("--tlen", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("C4_BENCH_FORCE_DIST") == "1"     # the latter: exercise RCCL on 1 GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    import exonerate_amd as ex
    from exonerate_amd import workloads

    eng = ex.Engine(local_rank)
    model = ex.Model("est2genome")
    # shard-by-query: rank r owns pairs [r*B, (r+1)*B)
    pairs = workloads.est2genome_pairs(args.pairs, args.qlen, args.tlen, first=rank * args.pairs)
    batch = ex.ResidentBatch(eng, model, pairs)          # upload + residue coding + splice arrays: untimed
    first_pass_cells = sum((len(q) + 1) * (len(t) + 1) for q, t in pairs)
    batch.kernel_stats(ex.MODE_FIND_REGION, reset=True)  # switches HIP-event timing of the kernels on

    def flush_c_stdio():
        # RCCL prints a version banner through C stdio; push it out now so that the JSON line is the last
        # thing on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    flush_c_stdio()
    for _ in range(args.warmup):
        batch.run(2)
    for m in range(4):
        batch.kernel_stats(m, reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run(2)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    out = None
    stats = {m: batch.kernel_stats(m) for m in range(4)}
    n_aligned = sum(1 for i in range(min(args.pairs, 64)) if batch.alignment(i) is not None)
    if rank == 0:
        total_cells = first_pass_cells * world * args.steps
        value = total_cells / elapsed
        reg = stats[ex.MODE_FIND_REGION]
        # algorithmic bytes of one region-pass launch: per pair Q + T residue bytes,
        # 4 splice arrays x 4 B x T, 32 B of result
        algo_bytes = sum(len(q) + len(t) + 16 * len(t) + 32 for q, t in pairs)
        avg_ms = reg["ms"] / max(1, reg["launches"])
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch of the same kernel from the PMC passes committed under profiles/ (bench.py
        # itself cannot read PMCs); only quoted when the run has the configuration that was profiled
        traffic, kname, valu = None, "viterbi_kernel_mw<Est2GenomeDesc, MODE_REGION>", None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            if tj["config"] == {"pairs_per_gpu": args.pairs, "query_len": args.qlen, "target_len": args.tlen}:
                traffic, kname, valu = tj["bytes_per_launch"], tj["kernel"], tj.get("valu")
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "DP cells/s (first-pass lattice cells / end-to-end time), est2genome 1kb x 100kb batch, "
                      "bit-exact vulgar vs reference",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "alignments_per_s": args.pairs * world * args.steps / elapsed,
            "config": {"workload": "est2genome (exhaustive Optimal_find_path, -D 32, --revcomp no), %d cDNAs of "
                                   "%d nt x genomic windows of %d nt per GPU, shard-by-query"
                                   % (args.pairs, args.qlen, args.tlen),
                       "pairs_per_gpu": args.pairs, "query_len": args.qlen, "target_len": args.tlen,
                       "aligned_in_sample": n_aligned},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": algo_bytes,
                         "kernel": kname, "valu_pmc": valu,
                         "avg_launch_ms": avg_ms, "launches": reg["launches"],
                         "kernel_cells_per_s": reg["cells"] / (reg["ms"] * 1e-3) if reg["ms"] else 0.0,
                         "note": "integer max-plus with all live DP state in VGPRs: compulsory HBM traffic is "
                                 "~17 B per target column, so the kernel is VALU-bound by construction; "
                                 "valu_pmc (profiles/) is the SQ-counter view of that bound (DESIGN.md section 5)"},
            "kernel_ms": {"region": stats[2]["ms"], "checkpoint": stats[3]["ms"], "path": stats[1]["ms"]},
        }
        # the CPU legs run at N=1 only: at N>1 the other ranks would sit in the process-group teardown while
        # rank 0 times a host program
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, rank, model, pairs, batch, eng)
            out["speedup_vs_cpu_1core"] = value / out["cpu_baseline"]["value"] / world
            if out["cpu_baseline"]["kind"] == "reference":
                allc = cpu_baseline_all_cores(pairs, batch)
                if allc:
                    out["cpu_baseline_all_cores"] = allc
                    out["speedup_vs_cpu_all_cores"] = value / allc["value"] / world
    batch.close()
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
