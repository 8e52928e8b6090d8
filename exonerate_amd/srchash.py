"""One number for "the device code this library was built from": SHA-256 over the sources under exonerate_amd/csrc (names and
contents, sorted).  tools/summarise_profile.py writes it into profiles/traffic_latest.json beside the counters it summarises;
bench.py quotes those counters only while the tree it runs from still has that hash."""
import hashlib
import os


def csrc_hash(root=None):
    root = root or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    names = []
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith((".h", ".hip", ".inc", ".cc")):
                names.append(os.path.join(d, f))
    for p in sorted(names):
        h.update(os.path.relpath(p, root).encode() + b"\0")
        h.update(open(p, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
