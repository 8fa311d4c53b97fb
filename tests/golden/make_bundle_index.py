"""Extract the directory of the reference's bundled checkpoint index
(/root/reference/nmrgnn/models/baseline/variables/variables.index; the data shard is missing
upstream) into tests/golden/bundle_index.json: every (key, BundleEntryProto bytes) pair in table
order, plus the file's length and SHA-256 so the table writer can be checked byte for byte.
Uses its own minimal block walker (not nmrgnn_amd.tfbundle) so the fixture is independent of the
code under test.  Run in the build container only; the GPU box never sees /root/reference."""
import hashlib
import json
import os
import struct

SRC = "/root/reference/nmrgnn/models/baseline/variables/variables.index"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bundle_index.json")


def varint(b, i):
    r = s = 0
    while True:
        c = b[i]; i += 1
        r |= (c & 0x7F) << s; s += 7
        if c < 0x80:
            return r, i


def block(d, off, size):
    b = d[off:off + size]
    nr = struct.unpack("<I", b[-4:])[0]
    end = len(b) - 4 - 4 * nr
    i, key, out = 0, b"", []
    while i < end:
        sh, i = varint(b, i); ns, i = varint(b, i); vl, i = varint(b, i)
        key = key[:sh] + b[i:i + ns]; i += ns
        out.append((key, b[i:i + vl])); i += vl
    return out


def main():
    d = open(SRC, "rb").read()
    foot = d[-48:]
    _, i = varint(foot, 0); _, i = varint(foot, i)
    io, i = varint(foot, i); isz, i = varint(foot, i)
    items = []
    for _, h in block(d, io, isz):
        o, j = varint(h, 0); s, _ = varint(h, j)
        items += block(d, o, s)
    doc = {"source": "nmrgnn/models/baseline/variables/variables.index (reference bundle directory; no weight values)",
           "length": len(d), "sha256": hashlib.sha256(d).hexdigest(),
           "entries": [[k.decode(), v.hex()] for k, v in items]}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=0)
    print(len(items), "entries ->", OUT)


if __name__ == "__main__":
    main()
