#!/usr/bin/env python3
"""Per basic block of a gfx950 assembly listing (hipcc --offload-device-only -S): instruction count, packed adds, scratch
loads / stores, s_waitcnt, and whether the block branches back to itself or an earlier label: where a kernel's spills sit."""
import re, sys
kernel = None
blocks = []
cur = None
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        kernel = m.group(1); cur = None; continue
    m = re.match(r'^(\.LBB\d+_\d+):', line)
    if m:
        cur = {"k": kernel, "label": m.group(1), "n": 0, "pk": 0, "sl": 0, "ss": 0, "wait": 0, "back": ""}
        blocks.append(cur); continue
    if cur is None: continue
    t = line.strip()
    if not t or t.startswith(('.', ';', '//')): continue
    cur["n"] += 1
    if t.startswith(("v_pk_add_i16", "v_pk_max_i16", "v_pk_sub_i16")): cur["pk"] += 1
    if t.startswith("scratch_load"): cur["sl"] += 1
    if t.startswith("scratch_store"): cur["ss"] += 1
    if t.startswith("s_waitcnt"): cur["wait"] += 1
    m = re.match(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', t)
    if m: cur["back"] += " " + (m.group(1) or m.group(2))
want = sys.argv[2] if len(sys.argv) > 2 else ""
for b in blocks:
    if want in b["k"] and b["n"] >= int(sys.argv[3]) if len(sys.argv) > 3 else 200:
        print(b["k"][-60:], b["label"], "insts", b["n"], "pk", b["pk"], "scratch ld/st", b["sl"], b["ss"], "waits", b["wait"], "->", b["back"])
