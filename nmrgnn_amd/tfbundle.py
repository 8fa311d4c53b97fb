"""Reader / writer for TensorFlow checkpoint bundles (``variables.index`` + ``variables.data-*``).

The reference saves and loads its model as a Keras SavedModel (nmrgnn/library.py:92-103,
nmrgnn/main.py:87-90); the weights live in ``<model>/variables/variables.{index,data-00000-of-00001}``.
TensorFlow is not a dependency of this build, so the on-disk format is restated here:

* ``.index`` is a LevelDB-style sorted table: data blocks of prefix-compressed entries
  ``[shared varint][non_shared varint][value_len varint][key tail][value]`` with a restart point every
  16 entries, each block followed by a 1-byte compression type (0) and a masked CRC-32C; an empty
  metaindex block; an index block (separator key -> block handle); a 48-byte footer ending in the
  magic 0xdb4775248b80fb57.  The first entry has the empty key and holds ``BundleHeaderProto``
  {1:num_shards, 2:endianness, 3:version{1:producer}}; every other value is a ``BundleEntryProto``
  {1:dtype, 2:shape{2:dim{1:size}}, 3:shard_id, 4:offset, 5:size, 6:crc32c (fixed32, masked)}.
* ``.data-SSSSS-of-NNNNN`` holds the raw little-endian tensor bytes at [offset, offset+size).

Pinned by the reference's own bundle index (tests/golden/bundle_index.json, test_tfbundle.py): the
reader recovers its 93 entries, the writer regenerates the file byte for byte, and the CRC matches
the stored checksums of the scalars whose values are known (0.0, beta_1 = 0.9, ...).

Variable-name mapping for the GNN model: ``variables/i`` in creation order = edge-fc
Dense {kernel,bias} pairs, then the MPLayer ``w`` tensors (rank 3), then fc-block Dense pairs;
``out_layer/{kernel,bias}``; ``embed_layer/kernel``; Adam slots under ``.OPTIMIZER_SLOT/optimizer/{m,v}``.
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict, namedtuple

import numpy as np

MAGIC = 0xdb4775248b80fb57
RESTART_INTERVAL = 16
BLOCK_SIZE = 262144            # table::Options default used by BundleWriter
VALUE_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
HEADER_VALUE = b"\x08\x01\x1a\x02\x08\x01"      # num_shards=1, little endian (default), version.producer=1

DT_FLOAT, DT_DOUBLE, DT_INT32, DT_STRING, DT_INT64, DT_BOOL = 1, 2, 3, 7, 9, 10
_NP_OF = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"), DT_INT32: np.dtype("<i4"),
          DT_INT64: np.dtype("<i8"), DT_BOOL: np.dtype("bool"), 4: np.dtype("u1"), 6: np.dtype("i1"),
          19: np.dtype("<f2")}
_DT_OF = {np.dtype("float32"): DT_FLOAT, np.dtype("float64"): DT_DOUBLE, np.dtype("int32"): DT_INT32,
          np.dtype("int64"): DT_INT64, np.dtype("bool"): DT_BOOL}

Entry = namedtuple("Entry", "dtype shape shard offset size crc32c")


# ---------------------------------------------------------------------------------- CRC-32C
def _make_table():
    t = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        t[i] = c
    return t


_T = _make_table()
_TL = [int(x) for x in _T]


def _crc_raw(buf, c):
    for b in buf:
        c = _TL[(c ^ b) & 0xFF] ^ (c >> 8)
    return c


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _shift_operator(nbytes):
    """32x32 GF(2) matrix (as 32 column words) advancing a raw CRC state over ``nbytes`` zero bytes."""
    op = [0] * 32                      # one zero BIT: state -> (state >> 1) ^ (poly if state & 1)
    op[0] = 0x82F63B78
    for i in range(1, 32):
        op[i] = 1 << (i - 1)
    result = [1 << i for i in range(32)]
    nbits = nbytes * 8
    while nbits:
        if nbits & 1:
            result = [_gf2_times(op, col) for col in result]
        op = [_gf2_times(op, col) for col in op]
        nbits >>= 1
    return result


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of ``data`` continuing from ``crc``.  Large buffers are cut into equal
    lanes that numpy advances in lock step; lane results are merged with the zero-shift operator."""
    buf = memoryview(data).cast("B") if not isinstance(data, (bytes, bytearray)) else data
    n = len(buf)
    c = crc ^ 0xFFFFFFFF
    if n < 1 << 14:
        return _crc_raw(buf, c) ^ 0xFFFFFFFF
    lanes = 2048
    L = n // lanes
    body = np.frombuffer(buf, dtype=np.uint8, count=lanes * L).reshape(lanes, L)
    state = np.zeros(lanes, dtype=np.uint32)
    state[0] = c                         # the running state enters lane 0; other lanes start from 0 (linearity)
    cols = np.ascontiguousarray(body.T)
    for j in range(L):
        state = _T[(state ^ cols[j]) & 0xFF] ^ (state >> 8)
    op = _shift_operator(L)
    acc = 0
    for s in state:
        acc = _gf2_times(op, acc) ^ int(s)
    return _crc_raw(bytes(buf[lanes * L:]), acc) ^ 0xFFFFFFFF


def mask_crc(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - 0xa282ead8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------- protobuf bits
def _get_varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def _put_varint(v):
    out = bytearray()
    while True:
        if v < 0x80:
            out.append(v)
            return bytes(out)
        out.append((v & 0x7F) | 0x80)
        v >>= 7


def decode_entry(v):
    dtype = shard = offset = size = crc = 0
    shape = []
    i = 0
    while i < len(v):
        tag, i = _get_varint(v, i)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            x, i = _get_varint(v, i)
            if f == 1: dtype = x
            elif f == 3: shard = x
            elif f == 4: offset = x
            elif f == 5: size = x
        elif wt == 5:
            x = struct.unpack_from("<I", v, i)[0]
            i += 4
            if f == 6: crc = x
        elif wt == 2:
            ln, i = _get_varint(v, i)
            sub = v[i:i + ln]
            i += ln
            if f == 2:                              # TensorShapeProto: repeated dim{1:size}
                j = 0
                while j < len(sub):
                    t2, j = _get_varint(sub, j)
                    l2, j = _get_varint(sub, j)
                    d = sub[j:j + l2]
                    j += l2
                    if t2 >> 3 == 2:
                        sz = 0
                        if d:
                            _, k = _get_varint(d, 0)
                            sz, _ = _get_varint(d, k)
                        shape.append(sz)
        else:
            raise ValueError(f"BundleEntryProto: unsupported wire type {wt}")
    return Entry(dtype, tuple(shape), shard, offset, size, crc)


def encode_entry(e):
    out = bytearray()
    if e.dtype:
        out += b"\x08" + _put_varint(e.dtype)
    shp = bytearray()
    for d in e.shape:
        dim = (b"\x08" + _put_varint(d)) if d else b""
        shp += b"\x12" + _put_varint(len(dim)) + dim
    out += b"\x12" + _put_varint(len(shp)) + shp
    if e.shard:
        out += b"\x18" + _put_varint(e.shard)
    if e.offset:
        out += b"\x20" + _put_varint(e.offset)
    if e.size:
        out += b"\x28" + _put_varint(e.size)
    if e.crc32c:
        out += b"\x35" + struct.pack("<I", e.crc32c)
    return bytes(out)


# ---------------------------------------------------------------------------------- table format
def _read_block(d, off, size, verify):
    b = d[off:off + size]
    if len(b) != size or off + size + 5 > len(d):
        raise ValueError("truncated table block")
    if d[off + size] != 0:
        raise ValueError("compressed table blocks are not supported (bundles are written uncompressed)")
    if verify:
        want = struct.unpack_from("<I", d, off + size + 1)[0]
        if mask_crc(crc32c(d[off:off + size + 1])) != want:
            raise ValueError("table block checksum mismatch")
    nr = struct.unpack_from("<I", b, size - 4)[0]
    end = size - 4 - 4 * nr
    i, key, out = 0, b"", []
    while i < end:
        sh, i = _get_varint(b, i)
        ns, i = _get_varint(b, i)
        vl, i = _get_varint(b, i)
        key = key[:sh] + b[i:i + ns]
        i += ns
        out.append((key, b[i:i + vl]))
        i += vl
    return out


def read_table(data, verify=True):
    """All (key, value) pairs of a LevelDB-format table, in order."""
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != MAGIC:
        raise ValueError("not a TensorFlow bundle index (bad magic)")
    foot = data[-48:]
    _, i = _get_varint(foot, 0)
    _, i = _get_varint(foot, i)
    io, i = _get_varint(foot, i)
    isz, i = _get_varint(foot, i)
    out = []
    for _, handle in _read_block(data, io, isz, verify):
        o, j = _get_varint(handle, 0)
        s, _ = _get_varint(handle, j)
        out.extend(_read_block(data, o, s, verify))
    return out


def _build_block(items):
    buf, restarts, last = bytearray(), [], b""
    for n, (k, v) in enumerate(items):
        shared = 0
        if n % RESTART_INTERVAL == 0:
            restarts.append(len(buf))
        else:
            m = min(len(last), len(k))
            while shared < m and last[shared] == k[shared]:
                shared += 1
        buf += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v))
        buf += k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    buf += struct.pack(f"<{len(restarts)}I", *restarts) + struct.pack("<I", len(restarts))
    return bytes(buf)


def _short_successor(k):
    for i, c in enumerate(k):
        if c != 0xFF:
            return k[:i] + bytes([c + 1])
    return k


def _short_separator(a, b):
    m = min(len(a), len(b))
    i = 0
    while i < m and a[i] == b[i]:
        i += 1
    if i < m and a[i] < 0xFF and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def build_table(items):
    """Serialise sorted (key, value) pairs the way TensorFlow's TableBuilder does (no compression)."""
    out = bytearray()
    index = []

    def emit(block):
        handle = _put_varint(len(out)) + _put_varint(len(block))
        out.extend(block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return handle

    groups, cur, est = [], [], 0
    for k, v in items:
        cur.append((k, v))
        est += len(k) + len(v) + 3
        if est >= BLOCK_SIZE:
            groups.append(cur)
            cur, est = [], 0
    if cur or not groups:
        groups.append(cur)
    for g, grp in enumerate(groups):
        handle = emit(_build_block(grp))
        last = grp[-1][0] if grp else b""
        sep = _short_separator(last, groups[g + 1][0][0]) if g + 1 < len(groups) else _short_successor(last)
        index.append((sep, handle))
    meta = emit(_build_block([]))
    idx = emit(_build_block(index))
    foot = meta + idx
    out += foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC)
    return bytes(out)


# ---------------------------------------------------------------------------------- bundles
def read_index(path, verify=True):
    """``path`` = the ``.index`` file (or the bundle prefix).  Returns an ordered name -> Entry map."""
    if not path.endswith(".index"):
        path += ".index"
    with open(path, "rb") as f:
        data = f.read()
    items = read_table(data, verify)
    if not items or items[0][0] != b"":
        raise ValueError("bundle index has no header entry")
    out = OrderedDict()
    for k, v in items[1:]:
        out[k.decode()] = decode_entry(v)
    return out


def _shard_path(prefix, shard, nshards):
    return f"{prefix}.data-{shard:05d}-of-{nshards:05d}"


def read_bundle(prefix, names=None, verify=True):
    """Read tensors of a checkpoint bundle into numpy arrays (string tensors -> list of bytes)."""
    if prefix.endswith(".index"):
        prefix = prefix[:-6]
    entries = read_index(prefix, verify)
    nshards = max(e.shard for e in entries.values()) + 1 if entries else 1
    files, out = {}, OrderedDict()
    try:
        for name, e in entries.items():
            if names is not None and name not in names:
                continue
            if e.shard not in files:
                p = _shard_path(prefix, e.shard, nshards)
                if not os.path.exists(p):
                    raise FileNotFoundError(
                        f"{p}: the bundle's data shard is missing (the index alone holds no weight values)")
                files[e.shard] = open(p, "rb")
            f = files[e.shard]
            f.seek(e.offset)
            raw = f.read(e.size)
            if len(raw) != e.size:
                raise ValueError(f"{name}: data shard truncated")
            if e.dtype == DT_STRING:
                out[name] = _decode_strings(raw, int(np.prod(e.shape, dtype=np.int64)), e, verify, name)
                continue
            if verify and e.crc32c and mask_crc(crc32c(raw)) != e.crc32c:
                raise ValueError(f"{name}: tensor checksum mismatch")
            if e.dtype not in _NP_OF:
                raise ValueError(f"{name}: unsupported dtype enum {e.dtype}")
            out[name] = np.frombuffer(raw, dtype=_NP_OF[e.dtype]).reshape(e.shape).copy()
    finally:
        for f in files.values():
            f.close()
    return out


def _decode_strings(raw, n, e, verify, name):
    i, lens = 0, []
    for _ in range(n):
        ln, i = _get_varint(raw, i)
        lens.append(ln)
    i += 4                                         # masked crc of the length prefix
    out = []
    for ln in lens:
        out.append(bytes(raw[i:i + ln]))
        i += ln
    return out


def write_bundle(prefix, tensors):
    """Write ``{name: ndarray}`` as a one-shard bundle: ``prefix.index`` + ``prefix.data-00000-of-00001``."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items, off = [(b"", HEADER_VALUE)], 0
    with open(_shard_path(prefix, 0, 1), "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name])            # (ascontiguousarray would turn scalars into [1])
            if a.dtype not in _DT_OF:
                raise ValueError(f"{name}: unsupported dtype {a.dtype}")
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            f.write(raw)
            e = Entry(_DT_OF[a.dtype], tuple(a.shape), 0, off, len(raw), mask_crc(crc32c(raw)))
            items.append((name.encode(), encode_entry(e)))
            off += len(raw)
    with open(prefix + ".index", "wb") as f:
        f.write(build_table(items))


# ---------------------------------------------------------------------------------- model mapping
def _value(name):
    return name + VALUE_SUFFIX


def _slot(name, slot):
    return f"{name}/.OPTIMIZER_SLOT/optimizer/{slot}{VALUE_SUFFIX}"


def variable_names(hp):
    """Our parameter name -> checkpoint object path, for a model with hyper-parameters ``hp``."""
    Le, L, Lf = hp.get('edge_fc_layers'), hp.get('mp_layers'), hp.get('fc_layers')
    out, i = OrderedDict(), 0
    for t in range(Le):
        out[f"edge_fc/{t}/kernel"] = f"variables/{i}"
        out[f"edge_fc/{t}/bias"] = f"variables/{i + 1}"
        i += 2
    for l in range(L):
        out[f"mp/{l}/w"] = f"variables/{i}"
        i += 1
    for t in range(Lf):
        out[f"fc/{t}/kernel"] = f"variables/{i}"
        out[f"fc/{t}/bias"] = f"variables/{i + 1}"
        i += 2
    out["out/kernel"] = "out_layer/kernel"
    out["out/bias"] = "out_layer/bias"
    out["embed/kernel"] = "embed_layer/kernel"
    return out


def infer_hypers(entries):
    """Architecture of a GNN-model bundle from the shapes in its index (dict name -> Entry)."""
    shapes, i = [], 0
    while _value(f"variables/{i}") in entries:
        shapes.append(entries[_value(f"variables/{i}")].shape)
        i += 1
    if not shapes or _value("embed_layer/kernel") not in entries:
        raise ValueError("not a GNN-model bundle (no variables/0 or embed_layer/kernel)")
    first_mp = next((j for j, s in enumerate(shapes) if len(s) == 3), None)
    if first_mp is None or first_mp % 2:
        raise ValueError("cannot locate the MPLayer weights in the bundle")
    n_mp = 0
    while first_mp + n_mp < len(shapes) and len(shapes[first_mp + n_mp]) == 3:
        n_mp += 1
    rest = len(shapes) - first_mp - n_mp
    if rest % 2 or rest == 0:
        raise ValueError("fc-block variables are not kernel/bias pairs")
    F, _, E = shapes[first_mp]
    num_elem, F2 = entries[_value("embed_layer/kernel")].shape
    if F2 != F:
        raise ValueError("embed_layer/kernel does not match atom_feature_size")
    return {"atom_feature_size": int(F), "edge_feature_size": int(E), "edge_hidden_size": int(shapes[0][0]),
            "edge_fc_layers": first_mp // 2, "mp_layers": n_mp, "fc_layers": rest // 2}, int(num_elem)


def load_gnn_bundle(prefix, hp=None, with_optimizer=False, verify=True):
    """Read a GNN-model bundle -> (state_dict in our names, hypers dict, num_elem[, optimizer state])."""
    if prefix.endswith(".index"):
        prefix = prefix[:-6]
    entries = read_index(prefix, verify)
    arch, num_elem = infer_hypers(entries)
    names = variable_names(type("H", (), {"get": lambda self, k: arch[k]})())
    want = {_value(v): k for k, v in names.items()}
    slots = {}
    if with_optimizer:
        for k, v in names.items():
            for s in ("m", "v"):
                if _slot(v, s) in entries:
                    slots[_slot(v, s)] = (s, k)
        for s in ("iter", "learning_rate", "beta_1", "beta_2"):
            if _value(f"optimizer/{s}") in entries:
                slots[_value(f"optimizer/{s}")] = ("opt", s)
    got = read_bundle(prefix, names=set(want) | set(slots), verify=verify)
    state = OrderedDict((want[k], got[k].astype(np.float32)) for k in want)
    if not with_optimizer:
        return state, arch, num_elem
    opt = {"m": {}, "v": {}}
    for k, (kind, name) in slots.items():
        if kind == "opt":
            opt[name] = got[k].item()
        else:
            opt[kind][name] = got[k].astype(np.float32)
    return state, arch, num_elem, opt


def save_gnn_bundle(prefix, state, hp, optimizer=None):
    """Write our parameters under the reference bundle's variable names (and Adam slots if given)."""
    names = variable_names(hp)
    tensors = {}
    for k, v in names.items():
        tensors[_value(v)] = np.asarray(state[k], dtype=np.float32)
    if optimizer:
        for s in ("m", "v"):
            for k, a in optimizer.get(s, {}).items():
                tensors[_slot(names[k], s)] = np.asarray(a, dtype=np.float32)
        if "iter" in optimizer:
            tensors[_value("optimizer/iter")] = np.asarray(optimizer["iter"], dtype=np.int64)
        for s in ("learning_rate", "beta_1", "beta_2"):
            if s in optimizer:
                tensors[_value(f"optimizer/{s}")] = np.asarray(optimizer[s], dtype=np.float32)
    write_bundle(prefix, tensors)
