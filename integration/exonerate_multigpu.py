#!/usr/bin/env python3
"""One exonerate-gpu process per GPU, sharded by query — the reference's own multi-process scheme
(--querychunkid / --querychunktotal, src/program/exonerate.c:64-75, src/database/fastadb.c:146-163), with each
process pinned to its device by --gpudevice.  Query chunks are contiguous byte ranges of the query file taken in
order, so the chunks' outputs concatenated in chunk order are the single-process output: this launcher prints
them that way (the "Command line:" / "Hostname:" banner of chunk 1 only, one "-- completed" line at the end).

Exception: --bestn.  The reference then defers all output to GAM_report, which walks its per-query tree in query-id
(strcmp) order (gam.c:551-553), so a single process prints globally id-sorted blocks while the chunks print
id-sorted blocks per chunk.  The same alignments are reported (--bestn is per query, hence shard-local); the
concatenation is byte-identical to the single-process output only when the query file is already sorted by id.

    exonerate_multigpu.py --gpus 8 [--devices 0,1,..] [--exe PATH] -- -m est2genome -E yes q.fa t.fa

No data-path collective: alignments of different queries are independent."""
import argparse, os, subprocess, sys, tempfile

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--devices", default=None, help="comma-separated HIP ordinals, one per process (default 0..gpus-1)")
    ap.add_argument("--exe", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "exonerate-gpu"))
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    rest = a.rest[1:] if a.rest and a.rest[0] == "--" else a.rest
    devs = [int(x) for x in a.devices.split(",")] if a.devices else list(range(a.gpus))
    assert len(devs) == a.gpus, "--devices needs one ordinal per process"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs, files = [], []
    for r in range(a.gpus):
        f = tempfile.TemporaryFile()
        files.append(f)
        procs.append(subprocess.Popen([a.exe, "--gpudevice", str(devs[r]), "--querychunkid", str(r + 1),
                                       "--querychunktotal", str(a.gpus)] + rest, stdout=f, env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    out = sys.stdout.buffer
    completed = None
    for r, f in enumerate(files):
        f.seek(0)
        for line in f:
            if line.startswith(b"-- completed"):
                completed = line
            elif r and (line.startswith(b"Command line:") or line.startswith(b"Hostname:")):
                continue
            else:
                out.write(line)
    if completed:
        out.write(completed)
    return rc

if __name__ == "__main__":
    sys.exit(main())
