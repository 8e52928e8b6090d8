"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares; host-side
entry points (params, models, memory decisions, printers) work without a GPU; device entry points fail
loudly instead of falling back."""
import ctypes as C
import os, re
import pytest
from exonerate_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(c4gpu_[a-z_0-9]+|c4m_[a-z_0-9]+)\s*\(", text))


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = _declared("c4gpu.h") | _declared("c4m.h")
    bound = {name for name, _, _ in _abi.PROTOTYPES}
    assert declared <= bound, sorted(declared - bound)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.c4gpu_abi_version() == _abi.ABI_VERSION == 9


def test_switches_are_read_once_and_reloaded_on_request(lib, monkeypatch):
    """csrc/c4_config.h: the library reads its C4GPU_* switches once; c4gpu_config_reload reads them again and says how many it
    found set (the tests' hook -- exonerate_amd/_abi.py calls it whenever the process's variables changed); no source file of the
    engine calls getenv."""
    for k in [k for k in os.environ if k.startswith("C4GPU_")]:
        monkeypatch.delenv(k)
    assert lib.c4gpu_config_reload() == 0
    monkeypatch.setenv("C4GPU_TRACE", "1")
    monkeypatch.setenv("C4GPU_PK16", "0")
    monkeypatch.setenv("C4GPU_NOT_A_SWITCH", "1")           # (not in the table: not counted)
    assert lib.c4gpu_config_reload() == 2
    monkeypatch.delenv("C4GPU_TRACE")
    monkeypatch.delenv("C4GPU_PK16")
    assert lib.c4gpu_config_reload() == 0
    csrc = os.path.join(ROOT, "exonerate_amd", "csrc")
    users = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".inc", ".h", ".cc")) and "getenv(" in re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f)).read()):
            users.append(f)
    assert users == ["c4_config.h"], users


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert not lib.c4gpu_ctx_create(0)
    assert b"no CPU fallback" in lib.c4gpu_last_error()
    import exonerate_amd as ex
    with pytest.raises(ex.C4GpuError):
        ex.Engine(0)


def test_product_does_not_touch_the_oracle():
    """Nothing under exonerate_amd/ may import, link or call oracle/ (the checker)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "exonerate_amd")):
        for f in files:
            if f.endswith((".py", ".cc", ".h", ".hip", ".inc")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("oracle_lib", "c4_oracle", "libc4oracle", "import oracle", "from oracle",
                               "oracle_find", "oracle_viterbi", "oracle/"):
                    assert needle not in text, (os.path.join(dirpath, f), needle)


def test_memory_decisions_match_the_oracle(lib, params):
    import oracle_lib
    from golden_util import get_model
    olib = oracle_lib.load()
    for name in ("affine_local_dna", "est2genome", "protein2dna"):
        m = get_model(lib, params, name)
        for (q, t) in ((5, 5), (6, 1000), (7, 13), (300, 300), (1000, 1000), (1000, 100000), (500, 1000000),
                       (40000, 60000)):
            for d in (0, 1, 32, 512):
                r = _abi.Region(0, 0, q, t)
                assert lib.c4gpu_use_reduced_space(m, r, d) == olib.oracle_use_reduced_space(m, r, d), (name, q, t, d)
                if lib.c4gpu_use_reduced_space(m, r, d):
                    assert lib.c4gpu_checkpoint_rows(m, r, d) == olib.oracle_checkpoint_rows(m, r, d), (name, q, t, d)


def test_printers_match_the_oracle(lib, params):
    """c4gpu_alignment_format == oracle formatter == the reference's lines, incl. reverse-strand coordinates."""
    import oracle_lib
    from golden_util import load_set, get_model
    olib = oracle_lib.load()
    for name in ("est2genome", "protein2dna", "affine_global_dna"):
        m = get_model(lib, params, name)
        for rec in load_set(name)[:12]:
            if "ops" not in rec:
                continue
            a = _abi.Alignment()
            a.score = rec["path_score"]
            a.region = _abi.Region(*rec["region"])
            n = len(rec["ops"])
            tr = (C.c_int32 * max(1, n))(*[o[0] for o in rec["ops"]])
            ln = (C.c_int32 * max(1, n))(*[o[1] for o in rec["ops"]])
            a.n_ops, a.op_transition, a.op_length = n, C.cast(tr, C.POINTER(C.c_int32)), C.cast(ln, C.POINTER(C.c_int32))
            for what, key in ((0, "sugar"), (1, "cigar"), (2, "vulgar")):
                b1, b2 = C.create_string_buffer(1 << 14), C.create_string_buffer(1 << 14)
                args = (rec["id"].encode(), rec["qlen"], b"+", b"tg", rec["tlen"], b"+", 1)
                assert lib.c4gpu_alignment_format(m, a, what, *args, b1, len(b1)) >= 0
                assert b1.value.decode() == rec[key]
                for strands in ((b"-", b"+"), (b"+", b"-")):
                    args = (rec["id"].encode(), rec["qlen"], strands[0], b"tg", rec["tlen"], strands[1], 1)
                    lib.c4gpu_alignment_format(m, a, what, *args, b1, len(b1))
                    olib.oracle_alignment_format(m, a, what, *args, b2, len(b2))
                    assert b1.value == b2.value


def test_subopt_point_sets_match_reference(lib, params):
    """c4gpu_subopt_add_alignment (host side, no device) against the blocked point set the reference itself
    held after each alignment of its sub-optimal loop (SubOpt_add_alignment, subopt.c:131)."""
    from golden_util import SUBOPT_SETS, load_set, get_model
    checked = 0
    for name in sorted(SUBOPT_SETS):
        model = get_model(lib, params, name)
        for rec in load_set(name):
            so = lib.c4gpu_subopt_create(len(rec["query"]), len(rec["target"]))
            for aln in rec["subopt"]:
                a = _abi.Alignment()
                a.score, a.region, a.n_ops, a.valid = aln["path_score"], _abi.Region(*aln["region"]), len(aln["ops"]), 1
                tr = (C.c_int32 * max(1, a.n_ops))(*[o[0] for o in aln["ops"]])
                ln = (C.c_int32 * max(1, a.n_ops))(*[o[1] for o in aln["ops"]])
                a.op_transition, a.op_length = C.cast(tr, C.POINTER(C.c_int32)), C.cast(ln, C.POINTER(C.c_int32))
                assert lib.c4gpu_subopt_add_alignment(so, model, a) == 0
                if "points" not in aln:
                    continue
                n = lib.c4gpu_subopt_points(so, None, None, 0)
                q, t = (C.c_int32 * max(1, n))(), (C.c_int32 * max(1, n))()
                lib.c4gpu_subopt_points(so, q, t, n)
                assert [[q[i], t[i]] for i in range(n)] == aln["points"], (name, rec["id"])
                checked += 1
            lib.c4gpu_subopt_destroy(so)
    assert checked > 50
