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
import subprocess
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


def reference_one_core(pair):
    """The reference's own compiled exonerate on one pair, one core: (cells/s record, its vulgar line), or None when the
    binary did not travel."""
    import subprocess, tempfile
    q, t = pair
    exe = os.path.join(ROOT, "oracle", "_ref", "exonerate-compiled")
    cells = (len(q) + 1) * (len(t) + 1)
    if not os.path.exists(exe) or cells > 3e8:
        return None
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
    except (OSError, subprocess.SubprocessError):
        return None
    lines = [l.strip() for l in r.stdout.decode().splitlines() if l.startswith("vulgar:")]
    if r.returncode != 0 or not lines:
        return None
    return ({"value": cells / cpu_s, "unit": "cells/s", "cores": 1, "kind": "reference",
             "sample": "pair 0 of the batch (%d x %d) through the reference's own exonerate (compiled Viterbi, -m est2genome "
                       "-E yes -S no --revcomp no), %.1f s wall incl. all passes, before the process group was formed"
                       % (len(q), len(t), cpu_s)}, lines[0])


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


class _StubBatch:
    """Stand-in for ResidentBatch in the CPU test of this file's control flow (C4_BENCH_STUB=1, gloo): no device,
    no alignment, fixed fake kernel statistics.  Never used for a measurement: the JSON line says "stub"."""

    def __init__(self, pairs):
        self.n = len(pairs)

    def run(self, what=2, **kw):
        time.sleep(0.002)

    def kernel_stats(self, mode, reset=False):
        return {"ms": 1.0, "launches": 1, "cells": 1000}

    def alignment(self, i):
        return None

    def export(self):
        import numpy as np
        head = np.zeros((self.n, 7), dtype=np.int32)
        head[:, 0] = 1; head[:, 1] = 100; head[:, 6] = 2
        return np.concatenate([head.ravel(), np.tile(np.array([11, 5, 23, 1], dtype=np.int32), self.n)])

    def swap(self, stage):
        self.n = stage.n

    def close(self):
        pass


class _StubStage:
    n = 0

    def load(self, pairs):
        time.sleep(0.001)
        self.n = len(pairs)
        return 1.0

    def close(self):
        pass


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU ourselves (the driver's
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` sets RANK/WORLD_SIZE and does not come
    here).  Rank 0's stdout is ours; the exit code is the worst of the ranks'."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def other_configs(ex, eng):
    """BASELINE.json's configurations 2, 3 and 5 at their full sizes on this GPU (exonerate_amd.workloads.bench_config): one
    warm-up and one timed pass of Optimal_find_path (-D 32) each; cells/s on first-pass cells, the device time of each kind of
    kernel, and every 64th alignment checked — score, region and operations — against tests/golden/bench_configs.json
    (generated by tools/make_bench_golden.py from the CPU oracle; tests/test_bench_golden.py keeps the file honest).  The
    headline `value` is configuration 4; these are reported beside it so that every configuration's number is the driver's."""
    from exonerate_amd import workloads
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_configs.json")))
    names = {0: "score", 1: "path", 2: "region", 3: "checkpoint"}
    out = {}
    for name, label in (("c2", "config 2: affine:local, 4096 DNA pairs of 1 kb x 1 kb"),
                        ("c3", "config 3: protein2dna, 1024 proteins of 500 aa x one 1 Mb contig"),
                        ("c5", "config 5 (exhaustive shape): protein2genome, 256 proteins of 300 aa x one 10 Mb chromosome")):
        model_name, pairs, _ = workloads.bench_config(name)
        model = ex.Model(model_name)
        b = ex.ResidentBatch(eng, model, pairs)
        b.run(2)                                                          # warm-up
        for m in range(4):
            b.kernel_stats(m, reset=True)
        t0 = time.perf_counter()
        b.run(2)
        dt = time.perf_counter() - t0
        ks = {m: b.kernel_stats(m) for m in range(4)}
        cells = sum((len(q) + 1) * (len(t) + 1) for q, t in pairs)
        checked = 0
        for rec in want[name]["sample"]:
            a = b.alignment(rec["pair"])
            ok = a is not None and a.score == rec["score"] and list(a.region) == rec["region"] and [list(o) for o in a.ops] == rec["ops"]
            assert ok, "%s: pair %d differs from tests/golden/bench_configs.json" % (name, rec["pair"])
            checked += 1
        dom = max(range(4), key=lambda m_: ks[m_]["ms"])
        out[name] = {"workload": label, "pairs": len(pairs), "first_pass_cells": cells, "ms_per_pass": dt * 1e3,
                     "value": cells / dt, "unit": "cells/s", "alignments_per_s": len(pairs) / dt,
                     "kernel_ms": {names[m]: ks[m]["ms"] for m in range(4)},
                     "dominant_kernel": {"pass": names[dom], "ms": ks[dom]["ms"], "launches": ks[dom]["launches"]},
                     "checked": "%d alignments (every %dth pair: score, region, operations) equal to tests/golden/bench_configs.json (records made by the reference: refdump)"
                                % (checked, want[name]["every"])}
        b.close()
    return out


def c5_heuristic_leg():
    """BASELINE config 5's heuristic leg through the drop-in binary (integration/_build/exonerate-gpu: the reference's own
    objects with libc4gpu.so behind its seams -- word scan, HSP extension and SDP on the device): 256 proteins of 300 aa against
    one 10 Mb chromosome, -m protein2genome, default mode; wall time of the best of three runs after a first, cold one (all four in wall_s_runs; runs started back to back alternate between fast and slow: the first large allocation waits while the driver clears what the run before released, profiles/r05_c5_cold.md), every stdout compared by SHA-256 with
    what the reference binary printed for the same input (tests/golden/bench_c5_heuristic.json, tools/make_c5_heuristic_golden.py:
    the reference needs 66 s on one core of the GPU box, too long for the bench).  Skipped where the binary is not built."""
    import hashlib, tempfile
    from exonerate_amd import workloads
    exe = os.path.join(ROOT, "integration", "_build", "exonerate-gpu")
    gold = os.path.join(ROOT, "tests", "golden", "bench_c5_heuristic.json")
    if not (os.path.exists(exe) and os.path.exists(gold)):
        return {}
    want = json.load(open(gold))
    with tempfile.TemporaryDirectory() as d:
        qf, tf = workloads.write_c5_heuristic_input(d)
        env = dict(os.environ, C4GPU_VERBOSE="1")
        dt, r, runs, slow = 0.0, None, [], []
        for _ in range(4):              # the first run pays the cold start; wall_s = the best of the other three, their median beside it
            t0 = time.perf_counter()
            r = subprocess.run([exe] + want["args"] + [qf, tf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            runs.append(round(time.perf_counter() - t0, 3))
            assert r.returncode == 0, r.stderr.decode()[-800:]
            assert hashlib.sha256(r.stdout).hexdigest() == want["sha256"], "c5 heuristic leg: output differs from the reference's"
        dt = min(runs[1:])
        if dt > 4.0:
            # where the time went when a run is slow (seen: hipMalloc of the 65 GB stream arena taking seconds instead of 0.3 ms)
            rt = subprocess.run([exe] + want["args"] + [qf, tf], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                env=dict(env, C4GPU_TRACE="1"))
            slow = [l.strip() for l in rt.stderr.decode().splitlines() if "arena of" in l or "sdp flush" in l][:12]
    assert hashlib.sha256(r.stdout).hexdigest() == want["sha256"], "c5 heuristic leg: output differs from the reference's"
    served = [l.split("c4gpu ", 1)[1].strip() for l in r.stderr.decode().splitlines() if "c4gpu sdp:" in l or "c4gpu seed:" in l]
    return {"c5_heuristic": {"workload": "config 5 (heuristic leg): exonerate-gpu -m protein2genome, 256 proteins of 300 aa x one 10 Mb "
                                         "chromosome, seeding + SDP on the device", "wall_s": dt, "alignments": want["alignments"],
                             "wall_s_cold": runs[0], "wall_s_warm_median": sorted(runs[1:])[len(runs[1:]) // 2],
                             "reference_wall_s_one_core": {"value": want.get("reference_wall_s_one_core_build_container"),
                                                           "machine": "build container (not the GPU box; 66.3 s were measured once on a GPU box's host, profiles/r03_c5_heuristic.md)"},
                             "checked": "stdout (%d vulgar lines) SHA-256 equal to the reference binary's, tests/golden/bench_c5_heuristic.json"
                                        % want["alignments"], "device": served, "wall_s_runs": runs,
                             **({"slow_run_trace": slow} if slow else {})}}


def c4_dropin_leg():
    """BASELINE config 4 through the command line: integration/_build/exonerate-gpu (the reference's own objects with libc4gpu.so
    behind the batching seam in front of GAM_Result_exhaustive_create) on 64 cDNAs x 64 genomic windows, all against all = 4 096
    rectangles of 1 001 x 100 001 cells, -m est2genome -E yes -S no --revcomp no: wall time of the whole process (FASTA parsing,
    flattening, device, replay through the reference's printers) of a cold run and of the better of two warm ones.  Checked against
    the reference binary on the same files (tests/golden/bench_c4_dropin.json, tools/make_c4_dropin_golden.py): the reference needs
    ~30 s per chance alignment across a 100 kb window (34 h of one core for all 4 096), so its output exists for the first
    `checked_queries` queries only, and stdout must BEGIN with exactly those bytes (SHA-256 of the head; the query loop is the outer
    one) and hold one vulgar line per rectangle.  Skipped where the binary or the golden file is missing."""
    import hashlib, tempfile
    from exonerate_amd import workloads
    exe = os.path.join(ROOT, "integration", "_build", "exonerate-gpu")
    gold = os.path.join(ROOT, "tests", "golden", "bench_c4_dropin.json")
    if not (os.path.exists(exe) and os.path.exists(gold)):
        return {}
    want = json.load(open(gold))
    cells = want["queries"] * want["targets"] * 1001 * 100001
    with tempfile.TemporaryDirectory() as d:
        qf, tf = workloads.write_c4_dropin_input(d, want["queries"], want["targets"])
        env = dict(os.environ, C4GPU_VERBOSE="1")
        runs, r = [], None
        for _ in range(3):
            t0 = time.perf_counter()
            r = subprocess.run([exe] + want["args"] + [qf, tf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            runs.append(round(time.perf_counter() - t0, 3))
            assert r.returncode == 0, r.stderr.decode()[-800:]
            assert hashlib.sha256(r.stdout[:want["head_bytes"]]).hexdigest() == want["sha256_head"], "c4 drop-in leg: output differs from the reference's"
            assert r.stdout.count(b"vulgar:") == want["queries"] * want["targets"]
    flush = [l.split("c4gpu: ", 1)[1].strip() for l in r.stderr.decode().splitlines() if "c4gpu: flush of" in l or "c4gpu: batch of" in l]
    warm = min(runs[1:])
    return {"c4_dropin": {"workload": "config 4 through the command line: exonerate-gpu -m est2genome -E yes -S no --revcomp no, 64 cDNAs x 64 "
                                      "windows all against all (4 096 rectangles of 1 001 x 100 001 cells), one process",
                          "wall_s_cold": runs[0], "wall_s_warm": warm, "wall_s_runs": runs, "value": cells / warm, "value_cold": cells / runs[0],
                          "unit": "cells/s", "alignments": want["queries"] * want["targets"],
                          "checked": "the first %d bytes of stdout (%d vulgar lines: queries 0-%d against all %d windows) SHA-256 equal to the "
                                     "reference binary's, tests/golden/bench_c4_dropin.json; %d vulgar lines in all"
                                     % (want["head_bytes"], want["head_alignments"], want["checked_queries"] - 1, want["targets"],
                                        want["queries"] * want["targets"]),
                          "reference_s_per_query_one_core": {"value": want["reference_s_per_query_one_core_build_container"],
                                                             "machine": "build container (not the GPU box)"},
                          "device": flush}}


def revcomp(seq):
    return seq.translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))[::-1]


NOMINAL_VALU_G = 0.5 * 1024 * 2.4          # G wave-instructions/s: one wave64 VALU instruction per two cycles and SIMD, 1 024 SIMDs, 2.4 GHz


def roofline_block(valu, hbm, extra):
    """The line's `roofline`: the roof that binds these kernels is VALU issue, so that is what achieved /
    peak / frac are; the HBM figures north_star asks for stay beside them under `hbm` (frac ~0 by construction: all live DP
    state is in registers, the compulsory traffic is ~17 B per target COLUMN).  Without counter figures for this tree the VALU
    numbers are null and the block says why (`counters`)."""
    out = {"bound": "valu-issue", "achieved": valu["achieved"] if valu else None, "peak": valu["peak"] if valu else NOMINAL_VALU_G,
           "unit": "G wave-inst/s", "frac": valu["frac"] if valu else None,
           "traffic": hbm.get("traffic"), "hbm": hbm, "valu": valu,
           "note": "integer max-plus with the whole anti-diagonal in VGPRs: achieved = the dominant kernel's wave-instructions per "
                   "launch (its lane operations per cell from the SQ counters of the profiled run of the same source hash x the "
                   "cells this run's launches covered) / this run's launch time (HIP events); peak = 0.5 wave-instructions per "
                   "cycle and SIMD x 1 024 SIMDs x 2.4 GHz; `hbm`: algorithmic bytes per launch / the same time against 8 TB/s "
                   "(DESIGN.md section 5)"}
    out.update(extra)
    return out


def kernel_table(kernels_pmc, stats, n_pairs, launches_per_step):
    """Per pass of the step: this run's launch time (HIP events) beside the profiled run's instruction count for the same code,
    i.e. the fraction of the nominal VALU issue rate each kernel reaches.  kernels_pmc: profiles/traffic_latest.json `kernels`
    (wave-instructions per PAIR of the north-star batch, registers, waiting share), None without counter figures."""
    names = {0: "score", 2: "windows", 3: "checkpoint", 1: "path"}
    table = {}
    for mode, name in names.items():
        st = stats[mode]
        if not st["launches"]:
            continue
        ms = st["ms"] / st["launches"]
        row = {"ms_per_launch": ms, "launches": st["launches"]}
        pk = (kernels_pmc or {}).get(name)
        if pk and ms > 0:
            insts = pk["wave_insts_per_pair"] * n_pairs / launches_per_step
            row.update({"wave_insts_per_launch": insts, "valu_frac_of_nominal": insts / (ms * 1e-3) / (NOMINAL_VALU_G * 1e9),
                        "wait_frac_profiled": pk.get("wait_frac"), "vgprs_per_lane": pk.get("vgprs_per_lane"),
                        "waves_per_simd": pk.get("waves_per_simd"), "kernel": pk.get("kernel")})
        table[name] = row
    return table


class ClockSampler:
    """Shader clock and package power of the GPU while the timed steps run (the packed score pass reaches
    2.29 GHz in the microbenchmark and 1.70 GHz beside a second launch lane; is that the power cap?): a thread that reads the
    driver's sysfs files four times a second -- no subprocess, nothing on the device."""
    def __init__(self, period=0.25):
        import glob
        self.period, self.samples, self._stop, self._thread = period, [], None, None
        self.pci = None
        self.sclk = self.power = None
        # the card this process computes on: a box has eight, the process sees one -- found by its PCI address
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        try:
            import torch
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            mine = [c for c in cards if os.path.basename(os.path.realpath(os.path.dirname(c))) == addr]
            cards = mine or cards
            self.pci = addr if mine else None
        except Exception:
            self.pci = None
        for c in cards:
            self.sclk = c
            pw = glob.glob(os.path.join(os.path.dirname(c), "hwmon", "hwmon*", "power1_average")) + \
                 glob.glob(os.path.join(os.path.dirname(c), "hwmon", "hwmon*", "power1_input"))
            self.power = pw[0] if pw else None
            break

    def _read(self):
        mhz = watts = None
        try:
            for line in open(self.sclk):
                if "*" in line:
                    mhz = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError, TypeError):
            pass
        try:
            watts = float(open(self.power).read()) / 1e6
        except (OSError, ValueError, TypeError):
            pass
        return mhz, watts

    def __enter__(self):
        import threading
        if self.sclk is None:
            return self
        try:
            self.raw = open(self.sclk).read()[:160]
        except OSError:
            self.raw = None
        self._stop = threading.Event()
        def loop():
            while not self._stop.is_set():
                self.samples.append(self._read())
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
        return False

    def report(self):
        mhz = [a for a, _ in self.samples if a]
        w = [b for _, b in self.samples if b]
        if not mhz and not w:
            return {"samples": 0, "note": "no readable shader clock / power file under /sys/class/drm (not a GPU box?)"}
        return {"samples": len(self.samples), "period_s": self.period,
                "sclk_mhz_mean": sum(mhz) / len(mhz) if mhz else None, "sclk_mhz_min": min(mhz) if mhz else None,
                "sclk_mhz_max": max(mhz) if mhz else None,
                "power_w_mean": sum(w) / len(w) if w else None, "power_w_max": max(w) if w else None,
                "pci_address_matched": self.pci, "pp_dpm_sclk_as_read": getattr(self, "raw", None),
                "source": "%s, %s (read while the K timed steps of the headline ran)" % (self.sclk, self.power)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=int(os.environ.get("C4_BENCH_PAIRS", "4096")),
                    help="pairs per GPU and step (BASELINE: 4096); with --scaling strong: pairs per step of the whole job")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank aligns --pairs pairs per step; strong: the --pairs pairs of a step are cut into "
                         "WORLD_SIZE shards (BASELINE config 4: 4 096 cDNAs -> 512 per GPU at 8 GPUs)")
    ap.add_argument("--qlen", type=int, default=1000)
    ap.add_argument("--tlen", type=int, default=100000)
    ap.add_argument("--distinct-batches", type=int, default=4,
                    help="different synthetic batches the steps cycle through (each step stages and aligns the next one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-revcomp", action="store_true", help="skip the extra both-strands measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE's other configurations (the `configs` block)")
    args = ap.parse_args()

    stub = os.environ.get("C4_BENCH_STUB") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python bench.py --gpus N starts "
                 "them itself; or python -m torch.distributed.run --nproc-per-node N bench.py --gpus N)" % (args.gpus, world))
    strong = args.scaling == "strong"
    if strong and args.pairs % world:
        sys.exit("bench.py: --scaling strong needs --pairs (%d) to be a multiple of the ranks (%d)" % (args.pairs, world))
    n_local = args.pairs // world if strong else args.pairs          # pairs this rank aligns per step
    n_step = args.pairs if strong else args.pairs * world            # pairs of the whole job per step

    # The synthetic input, made before any device or process group exists (the generator forks worker processes): `nb`
    # different batches per rank; step k stages and aligns batch k mod nb.  shard-by-query: batch b of the job is
    # pairs [b * n_step, (b + 1) * n_step) of the seeded generator, rank r owns the r-th n_local of them.
    from exonerate_amd import workloads
    nb = max(1, min(args.distinct_batches, args.steps + args.warmup))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    g0 = time.perf_counter()
    host_batches = workloads.est2genome_batches([b * n_step + rank * n_local for b in range(nb)], n_local, args.qlen, args.tlen,
                                                workers=max(1, min(32, cores // max(1, world))))
    # robustness legs (single GPU, full run only): one batch of 1 100-nt cDNAs (five strips of 256 rows: off the staged packed
    # score pass, whose queries must fit the four strips of a workgroup)
    want_robust = (not stub) and world == 1 and (not args.no_configs) and n_local >= 1024 and args.qlen == 1000
    long_q = workloads.est2genome_batches([nb * n_step], n_local, 1100, args.tlen,
                                          workers=max(1, min(32, cores)))[0] if want_robust else None
    # ... and one of 2 048 cDNAs of 2 500 nt (seven strips of 384 rows: two super-strips of the staged form, kpk16j)
    longer_q = workloads.est2genome_batches([(nb + 1) * n_step], n_local // 2, 2500, args.tlen,
                                            workers=max(1, min(32, cores)))[0] if want_robust else None
    gen_s = time.perf_counter() - g0
    pairs = host_batches[0]
    first_pass_cells = sum((len(q) + 1) * (len(t) + 1) for q, t in pairs)       # the same for every batch: fixed lengths

    import torch
    use_dist = world > 1 or os.environ.get("C4_BENCH_FORCE_DIST") == "1"     # the latter: exercise RCCL on 1 GPU
    # At N > 1 the reference's CPU leg (rank 0, one core, pair 0 of the batch) runs BEFORE the process group exists: the
    # other ranks wait in the rendezvous instead of in a collective; its vulgar line is compared with the GPU's later.
    early_cpu = None
    if use_dist and rank == 0 and not stub and not args.no_cpu_baseline:
        early_cpu = reference_one_core(pairs[0])
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not stub:
        torch.cuda.set_device(local_rank)

    def sync():
        if not stub:
            torch.cuda.synchronize()

    if stub:
        ex = eng = model = None
        batch, stage = _StubBatch(pairs), _StubStage()
        MODE_REGION = 2
    else:
        import exonerate_amd as ex
        eng = ex.Engine(local_rank)
        model = ex.Model("est2genome")
        MODE_REGION = ex.MODE_FIND_REGION
        # the batch object keeps the engine, the launch lanes and their buffers for the whole run; the sequences of every step
        # arrive through the stage (c4gpu_stage, include/c4gpu.h)
        batch = ex.ResidentBatch(eng, model, pairs[:2])
        stage = ex.Stage(eng, model)
    # Staging = what a caller pays per batch before the first DP cell: flattening into the library's structures, gathering the
    # residues into page-locked memory, the copy over PCIe, residue coding, the four splice-score arrays and the packed passes'
    # splice array built on the device (the reference's per-pair Sequence_strncpy + Intron_Data splice prediction).  Here: its
    # wall time ALONE (nothing else on the device), median of three loads after every buffer has reached its size; in the timed
    # steps below it runs behind the alignment of the batch before.
    for b in range(min(nb, 2)):
        stage.load(host_batches[b]); batch.swap(stage)
    alone = []
    for b in range(3):
        sync()
        s0 = time.perf_counter()
        stage.load(host_batches[b % nb])
        sync()
        alone.append(time.perf_counter() - s0)
    staging_s = sorted(alone)[1]
    batch.kernel_stats(MODE_REGION, reset=True)  # switches HIP-event timing of the kernels on

    def flush_c_stdio():
        # RCCL prints a version banner through C stdio; push it out now so that the JSON line is the last
        # thing on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass

    def barrier():
        sync()
        if use_dist:
            dist.barrier()
        sync()

    cdev = None
    if use_dist:
        from exonerate_amd import parallel
        cdev = parallel.collective_device()
    work = {"delivered": 0, "gathered_ints": 0, "queue_s": 0.0, "align_s": 0.0, "gather_s": 0.0, "stage_wait_s": 0.0}
    per_rank = {}
    from concurrent.futures import ThreadPoolExecutor
    loader = ThreadPoolExecutor(max_workers=1)          # the one thread that stages the next batch (a persistent worker: starting
    loader.submit(lambda: None).result()                # a thread per step costs the main thread a GIL hand-over of ~5 ms)

    def one_step(b, nxt, n_items):
        """One pass of the hot path over this rank's NEXT batch, as a work-queue step: rank 0 broadcasts the job
        header and scatters the work items (global pair ids; every rank reads its own shard of the input, as each process of a
        sharded exonerate run reads its own chunk of the query file), every rank takes the batch its stage has loaded, starts
        staging the one after it (`nxt`) behind the alignment, aligns, and the results (score, region, operations of every pair:
        c4gpu_batch_export) are gathered — tensors over RCCL/xGMI."""
        c0 = time.perf_counter()
        if use_dist:
            head = torch.tensor([n_items, world, 2, 32] if rank == 0 else [0, 0, 0, 0], dtype=torch.int64, device=cdev)
            dist.broadcast(head, src=0)                                   # job header: items per rank, ranks, mode, -D
            mine = torch.empty(n_items, dtype=torch.int64, device=cdev)
            if rank == 0:
                ids = torch.arange(world * n_items, dtype=torch.int64, device=cdev)
                dist.scatter(mine, [ids[r * n_items:(r + 1) * n_items].contiguous() for r in range(world)], src=0)
            else:
                dist.scatter(mine, None, src=0)
            assert int(head[0].item()) == n_items and int(mine[0].item()) == rank * n_items, "work items do not match the resident shard"
        work["queue_s"] += time.perf_counter() - c0                       # header broadcast + work-item scatter (incl. waiting for rank 0)
        c0 = time.perf_counter()
        b.swap(stage)                                                     # the batch staged during the step before
        fut = loader.submit(stage.load, nxt) if nxt is not None else None
        b.run(2)
        flat = b.export()                                                 # host copy of this rank's results, one stream
        c1 = time.perf_counter()
        if fut is not None:
            fut.result()                                                  # the next batch is resident (normally long since)
        work["stage_wait_s"] += time.perf_counter() - c1
        work["align_s"] += time.perf_counter() - c0                       # this rank's own alignment work
        c0 = time.perf_counter()
        if use_dist:
            got = parallel.all_gather_ragged(torch.from_numpy(flat).to(cdev), cdev)
            if rank == 0:
                total = 0
                for g in got:
                    hd = g[:7 * n_items].view(n_items, 7)
                    assert int(g.numel()) == 7 * n_items + 2 * int(hd[:, 6].sum().item()), "truncated result stream"
                    total += int(hd[:, 0].sum().item())
                work["delivered"] += total
                work["gathered_ints"] += sum(int(g.numel()) for g in got)
        else:
            work["delivered"] += int(flat[:7 * n_items].reshape(n_items, 7)[:, 0].sum())
            work["gathered_ints"] += int(flat.size)
        work["gather_s"] += time.perf_counter() - c0                      # result gather (incl. waiting for the slowest rank)

    def timed(b, batches, steps, warmup, streaming=True):
        """W untimed + exactly K timed steps between barriers; max over ranks.  streaming: step k takes the batch staged during
        step k - 1 and stages batch k + 1 behind its own alignment (K loads inside the timed region: the first batch's load is
        outside, the load started in the last step is waited for inside); otherwise every step re-aligns the resident batch."""
        n_items = len(batches[0])
        nbb = len(batches)
        barrier()
        stage.load(batches[0])
        k = 0
        if not streaming:
            b.swap(stage)
        for _ in range(warmup):
            one_step(b, batches[(k + 1) % nbb], n_items) if streaming else resident_step(b, n_items)
            k += 1
        for m in range(4):
            b.kernel_stats(m, reset=True)
        work["delivered"] = work["gathered_ints"] = 0
        work["queue_s"] = work["align_s"] = work["gather_s"] = work["stage_wait_s"] = 0.0
        barrier()
        t0 = time.perf_counter()
        step_ms = []
        for _ in range(steps):
            c0 = time.perf_counter()
            one_step(b, batches[(k + 1) % nbb], n_items) if streaming else resident_step(b, n_items)
            step_ms.append((time.perf_counter() - c0) * 1e3)
            k += 1
        per_rank["rank0_step_ms"] = step_ms
        t_own = time.perf_counter() - t0                                  # before the closing barrier: this rank's own K steps
        barrier()
        el = time.perf_counter() - t0
        mine = [el, t_own, work["align_s"], work["queue_s"], work["gather_s"], work["stage_wait_s"]]
        rows = [mine]
        if use_dist:
            tmax = torch.tensor([el], dtype=torch.float64, device="cpu" if stub else "cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
            # every rank's own clock, for the reader of a scaling run: which rank was slow, and was it its alignment work,
            # the work-queue collectives or the wait for the others
            tl = [torch.zeros(len(mine), dtype=torch.float64, device="cpu" if stub else "cuda") for _ in range(world)]
            dist.all_gather(tl, torch.tensor(mine, dtype=torch.float64, device="cpu" if stub else "cuda"))
            rows = [[float(x) for x in t.tolist()] for t in tl]
        per_rank["ms_per_step"] = [r[1] / steps * 1e3 for r in rows]
        per_rank["align_ms_per_step"] = [r[2] / steps * 1e3 for r in rows]
        per_rank["queue_ms_per_step"] = [r[3] / steps * 1e3 for r in rows]
        per_rank["gather_ms_per_step"] = [r[4] / steps * 1e3 for r in rows]
        per_rank["stage_wait_ms_per_step"] = [r[5] / steps * 1e3 for r in rows]
        a = per_rank["align_ms_per_step"]
        per_rank["align_imbalance_max_over_min"] = max(a) / min(a) if min(a) > 0 else None
        return el

    def resident_step(b, n_items):
        c0 = time.perf_counter()
        b.run(2)
        flat = b.export()
        work["align_s"] += time.perf_counter() - c0
        work["delivered"] += int(flat[:7 * n_items].reshape(n_items, 7)[:, 0].sum())
        work["gathered_ints"] += int(flat.size)

    flush_c_stdio()
    with ClockSampler() as clock_sampler:
        elapsed = timed(batch, host_batches, args.steps, args.warmup)
    stats = {m: batch.kernel_stats(m) for m in range(4)}
    ranks_report = dict(per_rank)                                         # of the headline run (the legs below overwrite it)
    headline_work = dict(work)

    # the same steps with the batch resident (what `value` was until round 4): every distinct batch staged, aligned once
    # untimed and once timed, on this rank only; the difference to the streaming step is what staging costs when it runs
    # behind the alignment
    res_ms = []
    if not use_dist:
        for bb in range(min(nb, 4)):
            stage.load(host_batches[bb]); batch.swap(stage)
            batch.run(2)
            sync()
            c0 = time.perf_counter()
            batch.run(2); batch.export()
            res_ms.append((time.perf_counter() - c0) * 1e3)
    resident_ms = sum(res_ms) / len(res_ms) if res_ms else None

    # BASELINE config 4 as worded ("4 096 cDNAs ... 1 -> 8 MI355X shard-by-query": 512 per GPU at 8): the shard one rank of a
    # strong-scaled 8-GPU run aligns per step, on this GPU alone, streaming like the headline
    shard = None
    if not stub and not use_dist and not args.no_configs and n_local >= 4096:
        shard_batches = [hb[:512] for hb in host_batches]
        keep = dict(work)
        sh_el = timed(batch, shard_batches, 4, 2)
        sh_stats = {m: batch.kernel_stats(m) for m in range(4)}
        work.update(keep)
        cells512 = sum((len(q) + 1) * (len(t) + 1) for q, t in shard_batches[0])
        shard = {"workload": "the 512-pair shard one of 8 ranks aligns per step under --scaling strong (4 096 cDNAs of the step "
                             "cut into 8), streaming, on this GPU alone", "pairs": 512, "ms_per_step": sh_el / 4 * 1e3,
                 "value": cells512 * 4 / sh_el, "unit": "cells/s",
                 "kernel_ms_per_step": {"score": sh_stats[0]["ms"] / 4, "region": sh_stats[2]["ms"] / 4, "checkpoint": sh_stats[3]["ms"] / 4,
                                        "path": sh_stats[1]["ms"] / 4}}

    # Robustness: what the headline's synthetic alphabet and query length hide.  (a) the same batch with N, R, Y, K sprinkled
    # over the targets (8 residue codes: the staged packed score pass holds a query profile for six); (b) cDNAs of 1 100 nt
    # (five strips).  Resident, one warm-up and two timed passes each; pair 0 of each against the reference binary.
    robust = None
    if want_robust and not use_dist:
        import numpy as np
        robust = {}

        def leg(name, label, batch_pairs):
            stage.load(batch_pairs); batch.swap(stage)
            batch.run(2)
            for m in range(4):
                batch.kernel_stats(m, reset=True)
            sync()
            c0 = time.perf_counter()
            for _ in range(2):
                batch.run(2); batch.export()
            dt = (time.perf_counter() - c0) / 2
            ks = {m: batch.kernel_stats(m) for m in range(4)}
            cells = sum((len(q) + 1) * (len(t) + 1) for q, t in batch_pairs)
            rec = {"workload": label, "ms_per_step_resident": dt * 1e3, "value": cells / dt, "unit": "cells/s",
                   "kernel_ms_per_step": {"score": ks[0]["ms"] / 2, "region": ks[2]["ms"] / 2, "checkpoint": ks[3]["ms"] / 2, "path": ks[1]["ms"] / 2}}
            if not args.no_cpu_baseline:
                ref = reference_one_core(batch_pairs[0])
                if ref is not None:
                    got = batch.alignment(0)
                    assert got is not None and got.vulgar("qy", "tg") == ref[1], "%s: GPU vulgar differs from the reference" % name
                    rec["checked"] = "pair 0 against the reference binary: vulgar identical"
            robust[name] = rec

        rng = np.random.default_rng(20260940)
        noisy = []
        for q, t in host_batches[0]:
            a = np.frombuffer(t, dtype=np.uint8).copy()
            pos = rng.integers(0, len(a), size=len(a) // 300)
            a[pos] = np.frombuffer(b"NRYK", dtype=np.uint8)[rng.integers(0, 4, size=len(pos))]
            noisy.append((q, a.tobytes()))
        leg("c4_eight_codes", "the north-star batch with N, R, Y, K at 0.3 % of the target positions (8 residue codes)", noisy)
        del noisy
        leg("c4_query_1100", "cDNAs of 1 100 nt (five strips of 256 rows) against 100 kb windows", long_q)
        leg("c4_query_2500", "%d cDNAs of 2 500 nt against 100 kb windows" % len(longer_q), longer_q)

    # both strands ("once with revcomp on, doubling cells"): what the reference does for DNA queries by
    # default (fastapipe.c:42-44): each cDNA and its reverse complement against the same window; the windows are
    # shared buffers, so the device holds each once.  Streaming like the headline; one warm-up + two timed steps.
    rc = None
    if not args.no_revcomp and not stub:
        both_batches = []
        for hb in host_batches[:2]:
            both = []
            for q, t in hb:
                both.append((q, t))
                both.append((revcomp(q), t))
            both_batches.append(both)
        keep = dict(work)
        rc_steps = 2
        rc_el = timed(batch, both_batches, rc_steps, 1)
        work.update(keep)
        rc = {"value": 2 * first_pass_cells * world * rc_steps / rc_el, "unit": "cells/s", "ms_per_step": rc_el / rc_steps * 1e3,
              "rectangles_per_gpu": len(both_batches[0]),
              "aligned_in_sample": sum(1 for i in range(min(len(both_batches[0]), 64)) if batch.alignment(i) is not None),
              "note": "--revcomp yes: every cDNA on both strands (2 x the first-pass cells), streaming (staging inside), "
                      "%d timed steps" % rc_steps}
        del both_batches

    # the checks below read alignments of batch 0 (rank 0: pairs 0 .. of the generator): stage and align it once more
    stage.load(host_batches[0]); batch.swap(stage); batch.run(2)
    n_aligned = sum(1 for i in range(min(n_local, 64)) if batch.alignment(i) is not None)

    # results compared across ranks: every rank aligns the same probe pair (pair 0 of rank 0's shard) on its own device;
    # the streams must be identical
    probe_same = None
    if use_dist and not stub:
        probe = workloads.est2genome_pairs(1, args.qlen, args.tlen, first=0)
        pb = ex.ResidentBatch(eng, model, probe)
        pb.run(2)
        streams = parallel.all_gather_ragged(torch.from_numpy(pb.export()).to(cdev), cdev)
        pb.close()
        probe_same = all(torch.equal(streams[0], x) for x in streams[1:]) and int(streams[0][0].item()) == 1
        assert probe_same, "ranks disagree on the probe pair"

    out = None
    if rank == 0:
        total_cells = first_pass_cells * world * args.steps
        value = total_cells / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        # the dominant kernel of the step: the whole-rectangle pass (FIND_REGION in one pass, or the FIND_SCORE pass of
        # the two-pass form: DESIGN.md section 4), whichever mode took the most device time
        dom_mode = max((0, 2), key=lambda m_: stats[m_]["ms"])
        reg = stats[dom_mode]
        # algorithmic bytes of one whole-rectangle launch: per pair Q + T residue bytes,
        # 4 splice arrays x 4 B x T, 32 B of result
        algo_bytes = sum(len(q) + len(t) + 16 * len(t) + 32 for q, t in pairs)
        avg_ms = reg["ms"] / max(1, reg["launches"])
        # a large batch runs as two halves on two launch lanes (c4_engine_find_path.inc, find_path_lanes): two launches of this kernel
        # per step, each over half of the pairs, sharing the device while they overlap; the roofline is per LAUNCH
        launches_per_step = max(1, round(reg["launches"] / max(1, args.steps)))
        algo_bytes_per_launch = algo_bytes / launches_per_step
        achieved = algo_bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch of the same kernel from the PMC passes committed under profiles/ (bench.py
        # itself cannot read PMCs); only quoted when the run has the configuration that was profiled
        traffic, valu_pmc = None, None
        kname = ("viterbi16_kernel_mw<Est2GenomeDesc> (FIND_SCORE + column dumps, two jobs per lane in packed 16-bit halves)"
                 if dom_mode == 0 and os.environ.get("C4GPU_PK16", "1") != "0" else
                 "viterbi_kernel_mw<Est2GenomeDesc, %s>" % ("MODE_SCORE + column dumps" if dom_mode == 0 else "MODE_REGION"))
        #  ... and only while the device code is the code that was profiled: the summary names the hash of
        # exonerate_amd/csrc it was made from (exonerate_amd/srchash.py); another tree gets no counter figures at all
        pmc_note, kernels_pmc = None, None
        try:
            from exonerate_amd.srchash import csrc_hash
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            here = csrc_hash()
            if tj.get("csrc_hash") != here:
                pmc_note = ("profiles/traffic_latest.json was made from csrc %s, this tree is %s: no counter figures quoted "
                            "(tools/profile_round.sh + tools/summarise_profile.py make new ones)" % (tj.get("csrc_hash"), here))
            elif tj["config"] == {"pairs_per_gpu": n_local, "query_len": args.qlen, "target_len": args.tlen} \
                    and tj.get("mode", 2) == dom_mode:
                traffic, kname, valu_pmc, kernels_pmc = tj["bytes_per_launch"], tj["kernel"], tj.get("valu"), tj.get("kernels")
                pmc_note = "counters of %s (csrc %s)" % (tj.get("source"), here)
            else:
                pmc_note = "profiles/traffic_latest.json is of another configuration: no counter figures quoted"
        except (OSError, ValueError, KeyError, ImportError) as e:
            pmc_note = "no usable profiles/traffic_latest.json (%s)" % type(e).__name__
        # the bound that matters for this kernel: VALU issue.  Peak = the MEASURED issue rate of the kernel's own
        # instruction mix at its occupancy (tools/valu_issue_microbench.hip -> profiles/valu_issue_latest.json),
        # instructions per launch from the SQ counters of the profiled run (profiles/traffic_latest.json),
        # launch time from this run's HIP events.
        valu = None
        try:
            vj = json.load(open(os.path.join(ROOT, "profiles", "valu_issue_latest.json")))
            if valu_pmc and avg_ms > 0:
                # peak: the nominal issue rate (one wave64 VALU instruction per 2 cycles and SIMD at the 2.4 GHz peak clock);
                # beside it the rate the packed max-plus transition itself reaches at this kernel's occupancy and the clock
                # it sustains there (tools/inst_class_microbench.hip: 3.0 cycles per instruction at three waves per SIMD)
                peak_ipc = vj.get("nominal_wave_inst_per_clk_per_simd", vj["peak_wave_inst_per_clk_per_simd"])
                clk = vj["clock_ghz"] * 1e9
                peak = peak_ipc * vj["simds"] * clk                     # wave-instructions per second, whole chip
                # instructions of THIS run's launches: the kernel's lane operations per cell (a property of the code: the SQ
                # counters of the profiled run of the same source hash) x the cells this run's launches covered / 64 lanes;
                # the time is this run's own (HIP events)
                insts_here = valu_pmc["lane_ops_per_cell"] * (reg["cells"] / max(1, reg["launches"])) / 64.0
                ach = insts_here / (avg_ms * 1e-3)
                valu = {"bound": "valu-issue", "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-inst/s",
                        "frac": ach / peak, "peak_wave_inst_per_clk_per_simd": peak_ipc, "clock_ghz": vj["clock_ghz"],
                        "lane_ops_per_cell": valu_pmc["lane_ops_per_cell"], "wave_insts_per_launch": insts_here,
                        "wait_frac_of_wave_cycles_profiled": valu_pmc.get("wait_frac_of_wave_cycles"),
                        "waves_per_simd": valu_pmc.get("waves_per_simd"), "source": vj["source"]}
                w = str(int(round(valu_pmc.get("waves_per_simd", 0))))
                if w in vj.get("wave_inst_per_clk_per_simd_by_waves", {}):
                    mix = vj["wave_inst_per_clk_per_simd_by_waves"][w] * vj["simds"] * vj["effective_clock_ghz_by_waves"][w] * 1e9
                    valu["mix_rate_at_occupancy"] = {"waves_per_simd": int(w), "wave_inst_per_clk_per_simd": vj["wave_inst_per_clk_per_simd_by_waves"][w],
                                                     "clock_ghz": vj["effective_clock_ghz_by_waves"][w], "peak": mix / 1e9, "frac": ach / mix}
        except (OSError, ValueError, KeyError):
            pass
        hidden = None
        if resident_ms is not None and staging_s > 0:
            hidden = max(0.0, min(1.0, 1.0 - (ms_per_step - resident_ms) / (staging_s * 1e3)))
        out = {
            "metric": "DP cells/s (first-pass lattice cells / end-to-end time), est2genome 1kb x 100kb batch, "
                      "bit-exact vulgar vs reference",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32 (packed int16 where proven exact)",
            "dtype_note": "int32 as the reference (typedef gint C4_Score); the whole-rectangle score pass runs two jobs per lane in "
                          "saturating packed int16 halves where every score provably fits (bit-identical results; C4GPU_PK16=0 "
                          "keeps it in int32)", "data": "synthetic" if not stub else "stub (control-flow test, no device)",
            "alignments_per_s": n_step * args.steps / elapsed,
            # every timed step stages and aligns a batch it has not seen in the step before: flattening, gather into page-locked
            # memory, PCIe copy, residue coding, splice arrays (the reference's per-pair Sequence_strncpy + splice prediction) are
            # INSIDE `value`, behind the alignment of the batch before (c4gpu_stage)
            "staging": {"inside_value": True, "ms_alone": staging_s * 1e3, "ms_alone_runs": [x * 1e3 for x in alone],
                        "ms_per_step_resident": resident_ms, "ms_per_step_resident_runs": res_ms,
                        "value_resident": first_pass_cells / (resident_ms * 1e-3) if resident_ms else None,
                        "hidden_frac": hidden, "distinct_batches": nb,
                        "stage_wait_ms_per_step": headline_work["stage_wait_s"] / max(1, args.steps) * 1e3,
                        "input_generation_s": gen_s,
                        "note": "hidden_frac = 1 - (ms_per_step - ms_per_step_resident) / ms_alone; ms_per_step_resident: the same "
                                "batches aligned again while resident (no staging), this rank alone; stage_wait: time a step "
                                "waited for the next batch's load after its own alignment was done"},
            "staging_ms": staging_s * 1e3, "staging_hidden_frac": hidden,
            "config": {"workload": "est2genome (exhaustive Optimal_find_path, -D 32, --revcomp no), %d cDNAs of "
                                   "%d nt x genomic windows of %d nt per GPU and step, a fresh batch every step, shard-by-query"
                                   % (n_local, args.qlen, args.tlen),
                       "pairs_per_gpu": n_local, "pairs_per_step": n_step, "query_len": args.qlen, "target_len": args.tlen,
                       "aligned_in_sample": n_aligned},
            "roofline": roofline_block(valu, {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                                              "algorithmic_bytes": algo_bytes_per_launch},
                                       {"launches_per_step": launches_per_step, "kernel": kname, "valu_pmc": valu_pmc,
                                        "avg_launch_ms": avg_ms, "launches": reg["launches"],
                                        "kernel_cells_per_s": reg["cells"] / (reg["ms"] * 1e-3) if reg["ms"] else 0.0,
                                        "counters": pmc_note, "clock": clock_sampler.report(),
                                        "kernels": kernel_table(kernels_pmc, stats, n_local, launches_per_step)}),
            "kernel_ms": {"score": stats[0]["ms"], "region": stats[2]["ms"], "checkpoint": stats[3]["ms"], "path": stats[1]["ms"]},
            # the step as a work-queue step: job header broadcast + work-item scatter + result gather (tensors over RCCL
            # when there is a process group), all inside the timed region
            "work_queue": {"collectives": "broadcast + scatter + all_gather (torch.distributed, backend %s)"
                                          % (("gloo" if stub else "nccl = RCCL") if use_dist else "none: one process"),
                           "alignments_delivered_per_step": headline_work["delivered"] / max(1, args.steps),
                           "result_ints_per_step": headline_work["gathered_ints"] / max(1, args.steps),
                           "probe_pair_identical_on_all_ranks": probe_same},
            # every rank's own clock over the K timed steps (before the closing barrier), split into its alignment work, the
            # work-queue collectives in front of it and the result gather behind it (both include waiting for other ranks)
            "ranks": ranks_report,
        }
        if rc:
            out["revcomp"] = rc
        if not args.no_configs and not use_dist and not stub:
            out["configs"] = other_configs(ex, eng)
            if shard:
                shard["frac_of_4096_pair_rate"] = shard["value"] / value
                # what eight ranks would reach on BASELINE config 4 as worded (4 096 cDNAs per step cut into eight shards) if
                # each aligned its shard at this rate: the work-queue collectives aside, 8 x the fraction
                shard["projected_speedup_8gpu_strong"] = 8.0 * shard["frac_of_4096_pair_rate"]
                out["configs"]["c4_shard512"] = shard
            if robust:
                for rec in robust.values():
                    rec["frac_of_resident_headline"] = rec["value"] / (first_pass_cells / (resident_ms * 1e-3)) if resident_ms else None
                out["configs"].update(robust)
        # the all-cores leg runs at N=1 only; at N>1 the one-core leg ran before the process group was formed (early_cpu)
        if early_cpu is not None:
            rec, ref_line = early_cpu
            got = batch.alignment(0)
            assert got is not None and got.vulgar("qy", "tg") == ref_line, "GPU vulgar differs from the reference: %r vs %r" % (got and got.vulgar(), ref_line)
            rec["vulgar_identical_to_gpu"] = True
            out["cpu_baseline"] = rec
            out["speedup_vs_cpu_1core"] = value / rec["value"] / world
        if not args.no_cpu_baseline and not use_dist and not stub:
            out["cpu_baseline"] = cpu_baseline(args, rank, model, pairs, batch, eng)
            out["speedup_vs_cpu_1core"] = value / out["cpu_baseline"]["value"] / world
            if out["cpu_baseline"]["kind"] == "reference":
                allc = cpu_baseline_all_cores(pairs, batch)
                if allc:
                    out["cpu_baseline_all_cores"] = allc
                    out["speedup_vs_cpu_all_cores"] = value / allc["value"] / world
    loader.shutdown()
    batch.close()
    stage.close()
    if eng:
        eng.close()
    # config 5's heuristic leg and config 4 through the drop-in binary are processes of their own: run once this process has
    # given the device's memory back (beside a resident batch their arenas are allocated ten times more slowly)
    if rank == 0 and isinstance(locals().get("out"), dict) and "configs" in out:
        out["configs"].update(c4_dropin_leg())
        out["configs"].update(c5_heuristic_leg())
    if use_dist:
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
