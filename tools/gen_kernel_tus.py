#!/usr/bin/env python3
"""Write the per-kernel translation units exonerate_amd/csrc/kernels/k_*.hip and the lookup table
kernels/table.cc.  One TU per (family, mode, continuation, scope specialisation) so `make -j` compiles
them in parallel; each TU only instantiates the hand-written template of c4_viterbi_kernel.h."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "exonerate_amd", "csrc", "kernels")
FAMILIES = [("FAM_UNGAPPED", "UngappedDesc", "ungapped"), ("FAM_AFFINE", "AffineDesc", "affine"),
            ("FAM_EST2GENOME", "Est2GenomeDesc", "est2genome"), ("FAM_UNGAPPED_P2D", "UngappedP2DDesc", "ungapped_p2d"),
            ("FAM_PROTEIN2DNA", "Protein2DnaDesc", "protein2dna"),
            ("FAM_PROTEIN2GENOME", "Protein2GenomeDesc", "protein2genome")]
# BSDP's derived models (small CORNER-scoped DPs, thousands per pair): score and quadratic-space path only
DERIVED = [("FAM_AFFINE_%s", "Affine%sDesc", "affine_%s"), ("FAM_EST2GENOME_FWD_%s", "Est2GenomeFwd%sDesc", "est2genome_fwd_%s"),
           ("FAM_EST2GENOME_REV_%s", "Est2GenomeRev%sDesc", "est2genome_rev_%s"),
           ("FAM_PROTEIN2DNA_%s", "Protein2Dna%sDesc", "protein2dna_%s"),
           ("FAM_PROTEIN2GENOME_%s", "Protein2Genome%sDesc", "protein2genome_%s")]
DERIVED_FAMILIES = [(f % k.upper(), d % k, s_ % k.lower()) for f, d, s_ in DERIVED for k in ("Start", "End", "Join")]
DERIVED_VARIANTS = [("MODE_SCORE", 0, 0, 0), ("MODE_PATH", 0, 0, 0)]
# (mode, cont, local, pack)
VARIANTS = [("MODE_SCORE", 0, 1, 0), ("MODE_SCORE", 0, 0, 0), ("MODE_REGION", 0, 1, 1), ("MODE_REGION", 0, 1, 0),
            ("MODE_REGION", 0, 0, 1), ("MODE_REGION", 0, 0, 0),
            ("MODE_PATH", 0, 0, 0), ("MODE_PATH", 1, 0, 0), ("MODE_CKPT", 1, 0, 0),
            # cont + local: the continuation kernels without the row-0 validity mask (c4_viterbi_kernel.h, eval_cell)
            ("MODE_PATH", 1, 1, 0), ("MODE_CKPT", 1, 1, 0)]
# rows per lane: the region/checkpoint passes carry 2-3 extra ints per state, so they get fewer rows
R = {("est2genome", "MODE_REGION"): 4, ("est2genome", "MODE_CKPT"): 3, ("est2genome", "MODE_PATH"): 1, ("protein2dna", "MODE_REGION"): 2,
     ("protein2dna", "MODE_CKPT"): 2, ("protein2genome", "MODE_SCORE"): 2, ("protein2genome", "MODE_REGION"): 2,
     ("protein2genome", "MODE_PATH"): 2, ("protein2genome", "MODE_CKPT"): 2}
# waves per SIMD the register allocator is held to (measured: est2genome region R=4 at 2 waves/SIMD beats
# R=2/3 uncapped and R=4 at 1 wave/SIMD, profiles/r01_variants.md)
# est2genome FIND_SCORE: 174 registers uncapped (185 with the column dumps) = 2 waves/SIMD; held to 168 it runs 3 (a
# handful of scratch accesses outside the column loop)
# est2genome FIND_PATH (one row per lane): 140 registers uncapped; its column loop needs fewer than 80, and held to 80 (six waves per
# SIMD instead of the four of rounds 1-5) the loops stay free of scratch -- the pass waits half of its cycles, more waves hide it
# (round 6: path pass 5.5 -> 5.1 ms on the north-star launch, both strands +4 %)
CAP = {("est2genome", "MODE_REGION"): 2, ("est2genome", "MODE_CKPT"): 2, ("est2genome", "MODE_PATH"): 6,
       ("est2genome", "MODE_SCORE"): 3}
# the unpacked region-start form (targets too long for (query_start << bits) | target_start in 31 bits)
# carries one more int per state: fewer rows per lane keep it inside 256 registers without scratch
R_UNPACKED = {("est2genome", "MODE_REGION"): 2}
EXPERIMENTS = {}                     # id -> (family, mode, rows per lane, register cap)
# est2genome has no non-local form (est2genome.c:58); its general (all validity masks kept) score / region kernels
# are still built in the one-wave form: they take over when the scoring parameters are too large for the
# local-scope shortcuts (Engine::local_exact, c4_engine_launch.inc)
# rows per lane of the seeded region windows (the dump loads of their first columns cost registers)
SEED2_R = {"est2genome": 3}
SEED2_CAP2 = {}
SEED2_R2 = {}
ONLY_LOCAL = {"est2genome"}

os.makedirs(OUT, exist_ok=True)
_written = set()
_builtin_open = open


class _KeepIfSame:
    """File object that only touches the file on disk when its content changed (make rebuilds by mtime)."""

    def __init__(self, path):
        self.path, self.parts = path, []

    def write(self, text):
        self.parts.append(text)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        text = "".join(self.parts)
        _written.add(os.path.basename(self.path))
        try:
            with _builtin_open(self.path) as fh:
                if fh.read() == text:
                    return False
        except OSError:
            pass
        with _builtin_open(self.path, "w") as fh:
            fh.write(text)
        return False


def open(path, mode="r"):       # every write below goes through the content check
    return _KeepIfSame(path) if mode == "w" else _builtin_open(path, mode)


entries = []
for fam, desc, short in DERIVED_FAMILIES:
    for mode, cont, local, pack in DERIVED_VARIANTS:
        sym = "k_%s_%s" % (short, mode[5:].lower())
        with open(os.path.join(OUT, sym + ".hip"), "w") as fh:
            fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                     'C4K_DEFINE_KERNEL(%s, %s, 1, %s, false, false, false, 4, false)\n}\n' % (sym, desc, mode))
        entries.append((fam, mode, cont, local, pack, 0, sym, 0))
# est2genome's span models: src (END cells out) and dst (START cells in); the src tables also serve the
# CORNER/CORNER "src traceback" model (heuristic.c:469-472)
SPAN = []
for model, stem, which in (("est2genome", "Est2Genome", "Fwd"), ("est2genome", "Est2Genome", "Rev"),
                           ("protein2genome", "Protein2Genome", "Phase0"), ("protein2genome", "Protein2Genome", "Phase1"),
                           ("protein2genome", "Protein2Genome", "Phase2")):
    for role, kernels in (("Src", (("MODE_SCORE", 2), ("MODE_SCORE", 0), ("MODE_PATH", 0))), ("Dst", (("MODE_SCORE", 1), ("MODE_PATH", 1)))):
        fam = "FAM_%s_%s_SPAN_%s" % (model.upper(), which.upper(), role.upper())
        desc = "%s%sSpan%sDesc" % (stem, which, role)
        for mode, span in kernels:
            sym = "k_%s_%s_span_%s_%s%s" % (model, which.lower(), role.lower(), mode[5:].lower(), ("_span%d" % span) if span else "")
            with open(os.path.join(OUT, sym + ".hip"), "w") as fh:
                fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                         'C4K_DEFINE_KERNEL_SPAN(%s, %s, 1, %s, false, false, false, 4, false, %d)\n}\n' % (sym, desc, mode, span))
            SPAN.append((fam, mode, span, sym))
for fam, desc, short in FAMILIES:
    for mode, cont, local, pack in VARIANTS:
        r = R.get((short, mode), 4)
        if not pack:
            r = R_UNPACKED.get((short, mode), r)
        cap = CAP.get((short, mode), 1)
        sym = "k_%s_%s%s%s%s" % (short, mode[5:].lower(), "_cont" if cont else "", "_local" if local else "",
                                 "_pack" if pack else "")
        if (short, mode, cont, local) == ("est2genome", "MODE_PATH", 1, 1):
            # the sub-alignment pass of the device route: jobs that name their alignment's root run the kernel of that root's
            # component (one strand: half the transitions), c4_viterbi_kernel.h BYROOT
            targs = "%s, %d, %s, true, true, %s, %d, false, 0, true" % (desc, r, mode, "true" if pack else "false", cap)
            with open(os.path.join(OUT, sym + ".hip"), "w") as fh:
                fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                         'static hipError_t %s_launch(const LaunchArgs &a) {\n'
                         '    hipLaunchKernelGGL((viterbi_kernel<%s>), dim3(a.grid), dim3(64), 0, a.stream, a.kp, a.seqs, a.jobs, a.n_jobs,\n'
                         '                       a.results, a.vsas, a.ops, a.scratch, a.queue);\n'
                         '    return hipGetLastError();\n}\n'
                         'const KernelInfo *%s() {\n'
                         '    static const KernelInfo ki = {%s_launch, (const void *)viterbi_kernel<%s>, "%s", %d,\n'
                         '                                  WaveDP<%s, %d, %s, true, true, false>::CS, WaveDP<%s, %d, %s, true, true, false>::BND,\n'
                         '                                  %s::NS, %s::MAXAT, 1, 0};\n'
                         '    return &ki;\n}\n}\n'
                         % (sym, targs, sym, sym, targs, sym, r, desc, r, mode, desc, r, mode, desc, desc))
        else:
          with open(os.path.join(OUT, sym + ".hip"), "w") as fh:
            fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                     'C4K_DEFINE_KERNEL(%s, %s, %d, %s, %s, %s, %s, %d, false)\n}\n'
                     % (sym, desc, r, mode, "true" if cont else "false", "true" if local else "false",
                        "true" if pack else "false", cap))
        entries.append((fam, mode, cont, local, pack, 0, sym, 0))
        if cont and local:            # the blocking kernels keep every mask
            continue
        # the same kernel with sub-optimal blocking (second and later alignments of a pair)
        ssym = sym + "_sub"
        with open(os.path.join(OUT, ssym + ".hip"), "w") as fh:
            fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                     'C4K_DEFINE_KERNEL(%s, %s, %d, %s, %s, %s, %s, %d, true)\n}\n'
                     % (ssym, desc, r, mode, "true" if cont else "false", "true" if local else "false",
                        "true" if pack else "false", cap))
        entries.append((fam, mode, cont, local, pack, 0, ssym, 1))
        # experiment variants: add (rows per lane, cap) pairs here and select them with C4GPU_WPE=<id>
        # at run time for an A/B on the GPU box (see profiles/r01_variants.md for the ones already measured)
        for vid, (vshort, vmode, rr, vcap) in EXPERIMENTS.items():
            if vshort == short and vmode == mode and (cont or local) and (pack or mode != "MODE_REGION"):
                wsym = "%s_v%d" % (sym, vid)
                with open(os.path.join(OUT, wsym + ".hip"), "w") as fh:
                    fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                             'C4K_DEFINE_KERNEL(%s, %s, %d, %s, %s, %s, %s, %d, false)\n}\n'
                             % (wsym, desc, rr, mode, "true" if cont else "false", "true" if local else "false",
                                "true" if pack else "false", vcap))
                entries.append((fam, mode, cont, local, pack, vid, wsym, 0))
# multi-wave (NW cooperating waves per job, LDS carry rings) kernels for the full-rectangle passes
MW = []
for fam, desc, short in FAMILIES:
    if short.startswith("ungapped"):
        continue
    for mode, cont, local, pack in VARIANTS:
        if cont or mode not in ("MODE_SCORE", "MODE_REGION"):
            continue
        if short in ONLY_LOCAL and not local:
            continue
        r = R.get((short, mode), 4)
        if not pack:
            r = R_UNPACKED.get((short, mode), r)
        cap = CAP.get((short, mode), 1)
        sym = "kmw_%s_%s%s%s" % (short, mode[5:].lower(), "_local" if local else "", "_pack" if pack else "")
        with open(os.path.join(OUT, sym + ".hip"), "w") as fh:
            fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                     'C4K_DEFINE_KERNEL_MW(%s, %s, %d, %s, %s, %s, 4, %d, false)\n}\n'
                     % (sym, desc, r, mode, "true" if local else "false", "true" if pack else "false", cap))
        MW.append((fam, mode, local, pack, sym, 4, 0, 0))
        # the windowed region pass (c4_viterbi_kernel.h, SEED): a score pass that dumps columns, a packed region pass
        # that starts from a dump
        if local and ((mode == "MODE_SCORE") or (mode == "MODE_REGION" and pack)):
            seed = 1 if mode == "MODE_SCORE" else 2
            dsym = "%s_seed%d" % (sym, seed)
            rs = SEED2_R.get(short, r) if seed == 2 else r
            with open(os.path.join(OUT, dsym + ".hip"), "w") as fh:
                fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                         'C4K_DEFINE_KERNEL_MW_SEED(%s, %s, %d, %s, true, %s, 4, %d, false, %d)\n}\n'
                         % (dsym, desc, rs, mode, "true" if pack else "false", cap, seed))
            MW.append((fam, mode, local, pack, dsym, 4, 0, seed))
            # the region windows on TWO cooperating waves (a window of a few hundred rows leaves fewer waves idle): measured
            # against the four-wave form with C4GPU_WIN_NW=2
            if seed == 2 and short == "est2genome":
                for nw2 in (2,):
                    d2 = dsym.replace("kmw_", "kmw%d_" % nw2)
                    with open(os.path.join(OUT, d2 + ".hip"), "w") as fh:
                        fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                                 'C4K_DEFINE_KERNEL_MW_SEED(%s, %s, %d, %s, true, %s, %d, %d, false, %d)\n}\n'
                                 % (d2, desc, SEED2_R2.get(short, rs), mode, "true" if pack else "false", nw2, SEED2_CAP2.get(short, cap), seed))
                    MW.append((fam, mode, local, pack, d2, nw2, 0, seed))
        # the sub-optimal rounds of the intron models run whole-rectangle passes too (score first, then region)
        if short == "est2genome" and local and (pack or mode == "MODE_SCORE"):
            ssym = sym + "_sub"
            with open(os.path.join(OUT, ssym + ".hip"), "w") as fh:
                fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                         'C4K_DEFINE_KERNEL_MW(%s, %s, %d, %s, %s, %s, 4, %d, true)\n}\n'
                         % (ssym, desc, r, mode, "true" if local else "false", "true" if pack else "false", cap))
            MW.append((fam, mode, local, pack, ssym, 4, 1, 0))
        # 8 waves x 2 rows per lane: the same 1 024 query rows per workgroup as 4 x 4, for launches with too
        # few jobs to fill the device and for targets whose HBM carry rows would not fit (one super-strip
        # needs none).  Only the intron models: their targets are the chromosome-sized ones.
        if short in ("est2genome", "protein2genome") and local:
            sym8 = sym.replace("kmw_", "kmw8_")
            with open(os.path.join(OUT, sym8 + ".hip"), "w") as fh:
                fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                         'C4K_DEFINE_KERNEL_MW(%s, %s, %d, %s, %s, %s, 8, %d, false)\n}\n'
                         % (sym8, desc, 2 if short == "est2genome" else 1, mode, "true" if local else "false",
                            "true" if pack else "false", 2 if short == "est2genome" else 1))
            MW.append((fam, mode, local, pack, sym8, 8, 0, 0))
            # ... and the score pass of the windowed region scheme in that shape (round 4): 256 proteins against one chromosome
            # are 256 jobs of three 128-row strips, less than one wave per SIMD, and a lone wave issues an instruction per
            # five cycles whatever it is; with one row per lane five waves of a job work (300 aa)
            if mode == "MODE_SCORE" and not pack:
                d8 = sym8 + "_seed1"
                with open(os.path.join(OUT, d8 + ".hip"), "w") as fh:
                    fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n'
                             'C4K_DEFINE_KERNEL_MW_SEED(%s, %s, %d, %s, true, false, 8, %d, false, 1)\n}\n'
                             % (d8, desc, 2 if short == "est2genome" else 1, mode, 2 if short == "est2genome" else 1))
                MW.append((fam, mode, local, pack, d8, 8, 0, 1))
with open(os.path.join(OUT, "table.cc"), "w") as fh:
    fh.write('// generated by tools/gen_kernel_tus.py\n#include "../c4_launch.h"\nnamespace c4k {\n')
    for e in entries:
        fh.write("const KernelInfo *%s();\n" % e[6])
    for e in SPAN:
        fh.write("const KernelInfo *%s();\n" % e[3])
    fh.write("const KernelInfo *get_kernel(int family, int mode, bool cont, bool local, bool pack, int wpe, bool sub, int span) {\n")
    for fam, mode, span, sym in SPAN:
        fh.write("    if (family == %s && mode == %s && !cont && !local && !pack && !sub && span == %d) return %s();\n" % (fam, mode, span, sym))
    fh.write("    if (span) return nullptr;\n")
    for fam, mode, cont, local, pack, wpe, sym, sub in sorted(entries, key=lambda e: -e[5]):
        fh.write("    if (family == %s && mode == %s && cont == %s && local == %s && pack == %s && sub == %s%s) return %s();\n"
                 % (fam, mode, "true" if cont else "false", "true" if local else "false",
                    "true" if pack else "false", "true" if sub else "false",
                    (" && wpe == %d" % wpe) if wpe and not sub else "", sym))
    fh.write("    return nullptr;\n}\n")
    for e in MW:
        fh.write("const KernelInfo *%s();\n" % e[4])
    fh.write("const KernelInfo *get_kernel_mw(int family, int mode, bool local, bool pack, int waves, bool sub, int seed) {\n")
    for fam, mode, local, pack, sym, nw, sub, seed in MW:
        fh.write("    if (family == %s && mode == %s && local == %s && pack == %s && waves == %d && sub == %s && seed == %d) return %s();\n"
                 % (fam, mode, "true" if local else "false", "true" if pack else "false", nw, "true" if sub else "false", seed, sym))
    fh.write("    return nullptr;\n}\n}\n")
for f in os.listdir(OUT):         # translation units of variants that no longer exist
    if f not in _written and not f.startswith(("ksdp_", "kpk16_", "kck16_", "kwin16_")):       # the SDP passes and the packed score pass: written by hand
        os.unlink(os.path.join(OUT, f))
print(len(entries), "kernels")
