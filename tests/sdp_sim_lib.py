"""ctypes binding of build/libc4sdpsim.so (tests/sdp_sim.hip): the product's sparse SDP wavefront — its per-lane cell
function, streams, walk and host side — driven by CPU loops.  TEST INFRASTRUCTURE: lets the build container (no GPU)
check the algorithm against the pinned oracle; never imported by exonerate_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle_lib
from exonerate_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "build", "libc4sdpsim.so")
SRC = [os.path.join(ROOT, "tests", "sdp_sim.hip")] + [os.path.join(ROOT, "exonerate_amd", "csrc", f)
                                                      for f in ("c4_sdp_wave.h", "c4_sdp_host.h", "c4_viterbi_kernel.h")]
FAMILY = {"affine": 1, "est2genome": 2, "protein2dna": 4, "protein2genome": 5}      # c4k::Family
_lib = None


def load():
    global _lib
    if _lib is None:
        if (not os.path.exists(SO)) or os.path.getmtime(SO) < max(os.path.getmtime(x) for x in SRC):
            os.makedirs(os.path.dirname(SO), exist_ok=True)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--offload-host-only", "-DC4SDP_HOST_SIM", "-O1", "-std=c++17",
                                   "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                                   "-I" + os.path.join(ROOT, "exonerate_amd", "csrc"), SRC[0], "-o", SO,
                                   "-L" + os.path.join(ROOT, "exonerate_amd"), "-lc4gpu",
                                   "-Wl,-rpath," + os.path.join(ROOT, "exonerate_amd")])
        _abi.load()
        lib = C.CDLL(SO)
        lib.sdpsim_pair.restype = C.c_int32
        lib.sdpsim_pair.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Params), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(_abi.Hsp), C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_abi.Alignment),
                                    C.POINTER(C.c_int64), C.c_int32]
        _lib = lib
    return _lib


def coded(params, family, q, t):
    """The arrays ResidentSeqs builds on the device (encode_kernel, codon_kernel, tn4_kernel, splice_kernel), with numpy
    and the oracle's splice restatement."""
    idx = np.frombuffer(bytes(params.submat_index), dtype=np.uint8)
    nt2d = np.frombuffer(bytes(params.nt2d), dtype=np.uint8).astype(np.int64)
    trans = np.frombuffer(bytes(params.trans), dtype=np.uint8)
    aa = np.frombuffer(bytes(params.aa), dtype=np.uint8)
    qa = np.frombuffer(q, dtype=np.uint8)
    ta = np.frombuffer(t, dtype=np.uint8)
    pad = 64
    qcode = np.zeros(len(q) + pad, dtype=np.uint8)
    qcode[:len(q)] = idx[qa]
    tcode = np.zeros(len(t) + pad, dtype=np.uint8)
    n = len(t)
    if family in ("protein2dna", "protein2genome"):
        if n >= 3:
            d = nt2d[ta]
            tcode[:n - 2] = idx[aa[trans[d[:-2] | (d[1:-1] << 4) | (d[2:] << 8)]]]
    else:
        tcode[:n] = idx[ta]
    assert qcode.max(initial=0) < 24 and tcode.max(initial=0) < 24
    ss = None
    if family in ("est2genome", "protein2genome"):
        ss = np.zeros((4, n + pad), dtype=np.int32)
        lib = oracle_lib.load()
        for k in range(4):
            buf = (C.c_int32 * max(1, n))()
            lib.oracle_splice_predict(C.byref(params.splice[k]), t, n, buf)
            ss[k, :n] = np.frombuffer(buf, dtype=np.int32)[:n]
    tn4 = None
    if family == "protein2genome":
        tn4 = np.zeros(n + pad, dtype=np.uint16)
        d = nt2d[ta]
        v = np.zeros(n, dtype=np.int64)
        for s in range(4):
            v[s:] |= d[:n - s] << (4 * s)
        tn4[:n] = v.astype(np.uint16)
    return qcode, tcode, ss, tn4


def sdp(model, params, family, q, t, hsps, query_advance=1, target_advance=1, dropoff=50, threshold=100, max_alignments=8,
        qid="qy", arena_chunks=2048):
    """(alignment dicts, wave steps executed): same contract as oracle_lib.sdp (single pass)."""
    lib = load()
    qcode, tcode, ss, tn4 = coded(params, family, q, t)
    n = len(hsps)
    hs = (_abi.Hsp * max(1, n))(*[_abi.Hsp(*h) for h in hsps])
    out = (_abi.Alignment * max_alignments)()
    steps = C.c_int64(0)
    k = lib.sdpsim_pair(model, params, FAMILY[family], qcode.ctypes.data, tcode.ctypes.data,
                        ss.ctypes.data if ss is not None else None, tn4.ctypes.data if tn4 is not None else None,
                        q, len(q), t, len(t), hs, n, query_advance, target_advance, dropoff, threshold, max_alignments, out,
                        C.byref(steps), arena_chunks)
    assert k >= 0, "sdpsim_pair failed: %d" % k
    res = []
    flib = _abi.load()
    for i in range(k):
        res.append(oracle_lib.alignment_to_dict(model, out[i], flib.c4gpu_alignment_format, qid, len(q), len(t)))
        flib.c4gpu_alignment_clear(out[i])
    return res, steps.value
