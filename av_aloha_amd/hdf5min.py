"""A minimal HDF5 writer / reader in pure Python + numpy, for the episode files of
data_collection_scripts/record_sim_episodes.py:186-206 (`h5py.File(..., 'w')`, `root.attrs['sim'] = True`, groups
`observations` / `observations/images`, float32 datasets (T, n), uint8 image datasets (T, H, W, 3) with chunks (1, H, W, 3)).
h5py is not installed where this build runs; the layout written here is the one h5py / libhdf5 write with their default
(`libver='earliest'`) settings, byte structures per the HDF5 File Format Specification 1.x:

  superblock version 0; object headers version 1; groups as symbol tables (B-tree v1 node type 0 + one symbol-table node + local
  heap); dataspace message v1, datatype message v1 (fixed point, IEEE float, the int8 enum {FALSE, TRUE} h5py uses for numpy bool),
  fill-value message v2, data layout message v3 (contiguous, or chunked with a B-tree v1 of node type 1, default K = 32, two levels
  when a data set has more than 64 chunks), attribute message v1.  No filters, no compression (the reference writes none either).

`write(path, datasets, attrs, chunks)` / `read(path) -> (datasets, attrs)`; names are absolute paths ("/observations/qpos").
The reader handles what the writer makes plus object-header continuation blocks and multi-node group trees, i.e. files h5py writes
with its defaults for this layout."""
from __future__ import annotations

import os
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"
GROUP_LEAF_K, GROUP_INTERNAL_K, CHUNK_K = 32, 16, 32      # symbol-table node holds 2 * leaf K entries; chunk trees 2 * 32 per node


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


# ---- datatype / dataspace / attribute encodings ----------------------------------------------------------------------------
def _dtype_msg(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt == np.bool_:               # h5py: enum over int8 with members FALSE = 0, TRUE = 1
        base = _dtype_msg(np.dtype("i1"))
        names = _pad8(b"FALSE\0") + _pad8(b"TRUE\0")
        return struct.pack("<BBBBI", 0x18, 2, 0, 0, 1) + base + names + bytes([0, 1])
    if dt.kind in "iu":
        bits0 = 0x08 if dt.kind == "i" else 0x00            # bit 3: signed; byte order little endian
        return struct.pack("<BBBBIHH", 0x10, bits0, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        if dt.itemsize == 4:
            sign, eloc, esz, msz, bias = 31, 23, 8, 23, 127
        else:
            sign, eloc, esz, msz, bias = 63, 52, 11, 52, 1023
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, sign, 0, dt.itemsize, 0, 8 * dt.itemsize, eloc, esz, 0, msz, bias)
    raise TypeError(f"hdf5min: unsupported dtype {dt}")


def _space_msg(shape) -> bytes:
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)


def _msg(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(msgs: list[bytes]) -> bytes:
    body = b"".join(msgs)
    return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body


def _attr_msg(name: str, value) -> bytes:
    a = np.asarray(value)
    nm = name.encode() + b"\0"
    dt, sp = _dtype_msg(a.dtype), _space_msg(a.shape)
    raw = a.astype(np.int8).tobytes() if a.dtype == np.bool_ else np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<")).tobytes()
    return struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + raw


# ---- writer ------------------------------------------------------------------------------------------------------------------
class _File:
    """Append-only writer straight to the file: a block is written where it is allocated (a data set's chunks from the array's own memory, no
    intermediate copy of the file in a bytearray -- an episode with six cameras is 1.9 GB), the few back-patches (superblock, B-tree siblings)
    seek."""
    def __init__(self, path: str):
        self.fh = open(path, "wb")
        self.pos = 0

    def alloc(self, data, pad_to: int = 0) -> int:
        """data: bytes, or a C-contiguous array (written through the buffer protocol); pad_to: zero-filled up to this many bytes."""
        gap = -self.pos % 8
        if gap:
            self.fh.write(b"\0" * gap)
            self.pos += gap
        addr = self.pos
        n = data.nbytes if isinstance(data, np.ndarray) else len(data)
        if n:
            self.fh.write(memoryview(data).cast("B") if isinstance(data, np.ndarray) else data)
        if pad_to > n:
            self.fh.write(b"\0" * (pad_to - n))
            n = pad_to
        self.pos += n
        return addr

    def patch(self, addr: int, data: bytes):
        self.fh.seek(addr)
        self.fh.write(data)
        self.fh.seek(self.pos)

    def close(self):
        self.fh.close()


def _chunk_tree(f: _File, arr: np.ndarray, chunk) -> int:
    """Writes the chunks of `arr` (only the first axis may be split: chunk = (c0, *arr.shape[1:])) and their B-tree; -> root address."""
    assert tuple(chunk[1:]) == tuple(arr.shape[1:]) and 1 <= chunk[0], "hdf5min: chunks may split the first axis only"
    c0 = int(chunk[0])
    nchunk = (arr.shape[0] + c0 - 1) // c0
    csize = c0 * int(np.prod(arr.shape[1:], dtype=np.int64)) * arr.itemsize
    assert csize < 2 ** 32
    entries = []                                     # (key offsets, child address)
    for k in range(nchunk):
        blk = arr[k * c0:(k + 1) * c0]
        entries.append((k * c0, f.alloc(np.ascontiguousarray(blk), pad_to=csize)))      # (a ragged last chunk is stored whole)
    return _chunk_index(f, arr.ndim, csize, entries, nchunk * c0)


def _chunk_index(f: _File, rank: int, csize: int, entries: list, end: int) -> int:
    """The chunk B-tree over chunks already in the file: entries = [(offset along the first axis, address)], `end` = the offset after the last."""
    def key(off0, size=csize):
        return struct.pack("<II", size, 0) + struct.pack("<Q", off0) + b"\0" * (8 * rank)      # rank + 1 offsets, the last (element) one 0
    klen = 8 + 8 * (rank + 1)
    node_bytes = 24 + 2 * CHUNK_K * 8 + (2 * CHUNK_K + 1) * klen

    def node(level, ents, last_off):
        b = b"TREE" + struct.pack("<BBHQQ", 1, level, len(ents), UNDEF, UNDEF)
        for off0, child in ents:
            b += key(off0) + struct.pack("<Q", child)
        b += key(last_off, 0)
        return b + b"\0" * (node_bytes - len(b))
    level = 0
    while True:
        groups = [entries[i:i + 2 * CHUNK_K] for i in range(0, len(entries), 2 * CHUNK_K)] or [[]]
        nxt = []
        for gi, g in enumerate(groups):
            last = groups[gi + 1][0][0] if gi + 1 < len(groups) else end
            nxt.append((g[0][0] if g else 0, f.alloc(node(level, g, last))))
        if len(nxt) == 1:
            return nxt[0][1]
        # sibling pointers are not needed for lookups; libhdf5 keeps them for iteration: fill them in
        for i, (_, a) in enumerate(nxt):
            f.patch(a + 8, struct.pack("<QQ", nxt[i - 1][1] if i else UNDEF, nxt[i + 1][1] if i + 1 < len(nxt) else UNDEF))
        entries, level = nxt, level + 1


class _Streamed:
    """A data set whose chunks (one per index of the first axis) were appended to the file as they came: StreamWriter.append."""
    def __init__(self, frame: np.ndarray):
        self.dtype, self.frame_shape, self.entries = frame.dtype, tuple(frame.shape), []


def _dataset(f: _File, arr, chunk) -> int:
    if isinstance(arr, _Streamed):
        shape, dtype = (len(arr.entries),) + arr.frame_shape, arr.dtype
        itemsize = np.dtype(dtype).itemsize
        chunk = (1,) + arr.frame_shape
        csize = int(np.prod(arr.frame_shape, dtype=np.int64)) * itemsize
        msgs = [_msg(0x0001, _space_msg(shape)), _msg(0x0003, _dtype_msg(np.dtype(dtype)), flags=1), _msg(0x0005, struct.pack("<BBBB", 2, 3, 2, 0))]
        root = _chunk_index(f, len(shape), csize, list(arr.entries), len(arr.entries))
        lay = struct.pack("<BBBQ", 3, 2, len(shape) + 1, root) + b"".join(struct.pack("<I", int(c)) for c in chunk) + struct.pack("<I", itemsize)
        msgs.append(_msg(0x0008, lay))
        return f.alloc(_object_header(msgs))
    arr = np.ascontiguousarray(arr)
    if arr.dtype.byteorder == ">":
        arr = arr.astype(arr.dtype.newbyteorder("<"))
    msgs = [_msg(0x0001, _space_msg(arr.shape)), _msg(0x0003, _dtype_msg(arr.dtype), flags=1),
            _msg(0x0005, struct.pack("<BBBB", 2, 3 if chunk else 2, 2, 0))]          # fill value v2: space allocation incremental (chunked) / late, written if set, none defined
    raw = arr.astype(np.int8) if arr.dtype == np.bool_ else arr
    if chunk:
        root = _chunk_tree(f, raw, chunk)
        lay = struct.pack("<BBBQ", 3, 2, arr.ndim + 1, root) + b"".join(struct.pack("<I", int(c)) for c in chunk) + struct.pack("<I", arr.itemsize)
    else:
        addr = f.alloc(np.ascontiguousarray(raw)) if arr.size else UNDEF
        lay = struct.pack("<BBQQ", 3, 1, addr, arr.nbytes)
    msgs.append(_msg(0x0008, lay))
    return f.alloc(_object_header(msgs))


def _group(f: _File, children: dict, attrs: dict | None = None):
    """children: name -> (object header address, (btree, heap) for groups | None).  -> (header address, btree, heap)"""
    names = sorted(children, key=lambda s: s.encode())
    assert len(names) <= 2 * GROUP_LEAF_K, "hdf5min: more links in one group than one symbol-table node holds"
    heap_data, offs = bytearray(b"\0" * 8), {}
    for n in names:
        offs[n] = len(heap_data)
        heap_data += _pad8(n.encode() + b"\0")
    heap_seg = f.alloc(bytes(heap_data))
    heap = f.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, heap_seg))      # free-list head 1 = none
    snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
    for n in names:
        oh, grp = children[n]
        if grp:
            snod += struct.pack("<QQII", offs[n], oh, 1, 0) + struct.pack("<QQ", *grp)
        else:
            snod += struct.pack("<QQII16x", offs[n], oh, 0, 0)
    snod += b"\0" * (8 + 2 * GROUP_LEAF_K * 40 - len(snod))
    snod_addr = f.alloc(snod)
    bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF)
    if names:
        bt += struct.pack("<QQQ", 0, snod_addr, offs[names[-1]])
    bt += b"\0" * (24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8 - len(bt))
    btree = f.alloc(bt)
    msgs = [_msg(0x0011, struct.pack("<QQ", btree, heap))] + [_msg(0x000C, _attr_msg(k, v)) for k, v in (attrs or {}).items()]
    return f.alloc(_object_header(msgs)), btree, heap


class StreamWriter:
    """A file written as its data arrives: `append(name, frame)` puts one more index of a chunked data set (chunk = one frame: the images of an
    episode, a step at a time, several data sets interleaved) straight into the file; `finish(datasets, attrs)` adds the whole data sets and the
    group structure.  Nothing of the streamed data is kept in memory but a chunk address per frame."""
    def __init__(self, path: str):
        self.path = path
        self.f = _File(path)
        self.f.alloc(b"\0" * 96)                             # superblock, patched at the end
        self.streams: dict = {}
        self.done = False

    def append(self, name: str, frame) -> None:
        frame = np.ascontiguousarray(frame)
        st = self.streams.get(name)
        if st is None:
            st = self.streams[name] = _Streamed(frame)
        assert frame.dtype == st.dtype and tuple(frame.shape) == st.frame_shape, f"hdf5min: a frame of {name} with another dtype or shape"
        st.entries.append((len(st.entries), self.f.alloc(frame)))

    def abort(self) -> None:
        """Closes and removes the file (an episode that is not kept)."""
        if not self.done:
            self.done = True
            self.f.close()
            try:
                os.remove(self.path)
            except OSError:
                pass

    def finish(self, datasets: dict | None = None, attrs: dict | None = None, chunks: dict | None = None) -> None:
        f, chunks = self.f, chunks or {}
        try:
            tree: dict = {}
            leaves = {**{k: (np.asarray(v), chunks.get(k)) for k, v in (datasets or {}).items()}, **{k: (v, None) for k, v in self.streams.items()}}
            for name, leaf in leaves.items():
                parts = [p for p in name.split("/") if p]
                node = tree
                for p in parts[:-1]:
                    node = node.setdefault(p, {})
                    assert isinstance(node, dict), f"hdf5min: {name} passes through a data set"
                node[parts[-1]] = leaf

            def build(node, top):
                ch = {}
                for k, v in node.items():
                    if isinstance(v, dict):
                        oh, bt, hp = build(v, False)
                        ch[k] = (oh, (bt, hp))
                    else:
                        ch[k] = (_dataset(f, v[0], v[1]), None)
                return _group(f, ch, attrs if top else None)
            root_oh, root_bt, root_hp = build(tree, True)
            eof = f.pos + (-f.pos % 8)
            if eof > f.pos:
                f.fh.write(b"\0" * (eof - f.pos))
                f.pos = eof
            sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
            sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
            sb += struct.pack("<QQII", 0, root_oh, 1, 0) + struct.pack("<QQ", root_bt, root_hp)
            assert len(sb) == 96
            f.patch(0, sb)
        except BaseException:
            self.abort()                 # no truncated file with a zeroed superblock behind a failed finish (bad name, dtype, disk full)
            raise
        self.done = True
        f.close()


def write(path: str, datasets: dict, attrs: dict | None = None, chunks: dict | None = None) -> None:
    """datasets: "/a/b/name" -> array; attrs: attributes of the root group; chunks: name -> chunk shape (first axis split only)."""
    # written next to the destination and renamed over it: a failure leaves an existing (good) file at `path` as it was
    part = path + ".part"
    StreamWriter(part).finish(datasets, attrs, chunks)
    os.replace(part, path)


# ---- reader ------------------------------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        if buf[:8] != SIG:
            raise ValueError("hdf5min: not an HDF5 file (signature at offset 0)")
        ver = buf[8]
        if ver > 1 or buf[13] != 8 or buf[14] != 8:
            raise NotImplementedError("hdf5min: superblock version 0 / 1 with 8-byte offsets and lengths only")
        self.leaf_k, self.int_k = struct.unpack_from("<HH", buf, 16)
        o = 24 + (4 if ver == 1 else 0)
        self.base, _, self.eof, _ = struct.unpack_from("<QQQQ", buf, o)
        self.root = struct.unpack_from("<QQII16s", buf, o + 32)

    def messages(self, addr):
        b = self.b
        ver, _, nmsg, _, size = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise NotImplementedError("hdf5min: object header version 1 only")
        blocks, out = [(addr + 16, size)], []
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                t, sz, fl = struct.unpack_from("<HHB", b, p)
                data = b[p + 8:p + 8 + sz]
                if t == 0x0010:
                    blocks.append(struct.unpack_from("<QQ", data))
                out.append((t, data))
                p += 8 + sz
        return out

    def dtype(self, d):
        cls, ver = d[0] & 15, d[0] >> 4
        size = struct.unpack_from("<I", d, 4)[0]
        if cls == 0:
            return np.dtype(("<" if not d[1] & 1 else ">") + ("i" if d[1] & 8 else "u") + str(size)), 8 + 4
        if cls == 1:
            return np.dtype(("<" if not d[1] & 1 else ">") + "f" + str(size)), 8 + 12
        if cls == 8:
            n = d[1] | (d[2] << 8)
            base, used = self.dtype(d[8:])
            p = 8 + used
            names = []
            for _ in range(n):
                e = d.index(b"\0", p)
                names.append(d[p:e].decode())
                p = p + ((e - p + 1 + 7) // 8) * 8 if ver < 3 else e + 1
            p += n * base.itemsize
            if sorted(names) == ["FALSE", "TRUE"] and base.itemsize == 1:
                return np.dtype(np.bool_), p
            return base, p
        raise NotImplementedError(f"hdf5min: datatype class {cls}")

    def space(self, d):
        ver, rank, fl = d[0], d[1], d[2]
        o = 8 if ver == 1 else 4
        return tuple(struct.unpack_from("<Q", d, o + 8 * i)[0] for i in range(rank))

    def heap_name(self, heap, off):
        seg = struct.unpack_from("<Q", self.b, heap + 24)[0]
        e = self.b.index(b"\0", seg + off)
        return self.b[seg + off:e].decode()

    def group_entries(self, btree, heap):
        b = self.b
        assert b[btree:btree + 4] == b"TREE"
        _, level, n = struct.unpack_from("<BBH", b, btree + 4)
        out = []
        for i in range(n):
            child = struct.unpack_from("<Q", b, btree + 24 + 8 + 16 * i)[0]
            if level > 0:
                out += self.group_entries(child, heap)
                continue
            assert b[child:child + 4] == b"SNOD"
            ns = struct.unpack_from("<H", b, child + 6)[0]
            for k in range(ns):
                noff, oh, ctype, _, scratch = struct.unpack_from("<QQII16s", b, child + 8 + 40 * k)
                out.append((self.heap_name(heap, noff), oh))
        return out

    def chunks(self, node, rank, out):
        b = self.b
        assert b[node:node + 4] == b"TREE"
        _, level, n = struct.unpack_from("<BBH", b, node + 4)
        klen = 8 + 8 * (rank + 1)
        for i in range(n):
            p = node + 24 + i * (klen + 8)
            size, mask = struct.unpack_from("<II", b, p)
            offs = struct.unpack_from("<" + "Q" * rank, b, p + 8)
            child = struct.unpack_from("<Q", b, p + klen)[0]
            if level > 0:
                self.chunks(child, rank, out)
            else:
                if mask:
                    raise NotImplementedError("hdf5min: filtered chunks")
                out.append((offs, child, size))

    def dataset(self, msgs):
        m = {t: d for t, d in msgs}
        dt, _ = self.dtype(m[0x0003])
        shape = self.space(m[0x0001])
        lay = m[0x0008]
        if lay[0] != 3:
            raise NotImplementedError("hdf5min: data layout message version 3 only")
        store = np.dtype("i1") if dt == np.bool_ else dt
        if lay[1] == 1:
            addr, size = struct.unpack_from("<QQ", lay, 2)
            a = np.frombuffer(self.b, dtype=store, count=int(np.prod(shape, dtype=np.int64)), offset=addr).reshape(shape) if size else np.zeros(shape, store)
        elif lay[1] == 2:
            rank = lay[2] - 1
            root = struct.unpack_from("<Q", lay, 3)[0]
            cdims = struct.unpack_from("<" + "I" * rank, lay, 11)
            a = np.zeros(shape, store)
            lst = []
            if root != UNDEF:
                self.chunks(root, rank, lst)
            for offs, addr, size in lst:
                blk = np.frombuffer(self.b, dtype=store, count=int(np.prod(cdims, dtype=np.int64)), offset=addr).reshape(cdims)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                a[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
        else:
            raise NotImplementedError("hdf5min: compact layout")
        return a.astype(np.bool_) if dt == np.bool_ else np.array(a)

    def attrs(self, msgs):
        out = {}
        for t, d in msgs:
            if t != 0x000C:
                continue
            ver, _, nsz, dsz, ssz = struct.unpack_from("<BBHHH", d)
            if ver != 1:
                raise NotImplementedError("hdf5min: attribute message version 1 only")
            p = 8
            name = d[p:p + nsz].split(b"\0")[0].decode(); p += (nsz + 7) // 8 * 8
            dt, _ = self.dtype(d[p:p + dsz]); p += (dsz + 7) // 8 * 8
            shape = self.space(d[p:p + ssz]); p += (ssz + 7) // 8 * 8
            store = np.dtype("i1") if dt == np.bool_ else dt
            a = np.frombuffer(d, dtype=store, count=int(np.prod(shape, dtype=np.int64)), offset=p).reshape(shape)
            a = a.astype(np.bool_) if dt == np.bool_ else np.array(a)
            out[name] = a[()] if a.shape == () else a
        return out

    def walk(self, oh, prefix, out):
        msgs = self.messages(oh)
        st = [d for t, d in msgs if t == 0x0011]
        if st:
            btree, heap = struct.unpack_from("<QQ", st[0])
            for name, child in self.group_entries(btree, heap):
                self.walk(child, prefix + "/" + name, out)
        elif any(t == 0x0008 for t, _ in msgs):
            out[prefix] = self.dataset(msgs)
        return msgs


def read(path: str):
    """-> ({"/group/name": array, ...}, {root attribute: value})"""
    with open(path, "rb") as fh:
        r = _Reader(fh.read())
    out: dict = {}
    msgs = r.walk(r.root[1], "", out)
    return out, r.attrs(msgs)


def layout(path: str) -> dict:
    """name -> ("contiguous", None) | ("chunked", chunk shape): what a reader of record_sim_episodes.py's files relies on."""
    with open(path, "rb") as fh:
        r = _Reader(fh.read())
    res = {}

    def walk(oh, prefix):
        msgs = r.messages(oh)
        st = [d for t, d in msgs if t == 0x0011]
        if st:
            for name, child in r.group_entries(*struct.unpack_from("<QQ", st[0])):
                walk(child, prefix + "/" + name)
        else:
            lay = dict(msgs).get(0x0008)
            if lay is not None:
                res[prefix] = ("chunked", struct.unpack_from("<" + "I" * (lay[2] - 1), lay, 11)) if lay[1] == 2 else ("contiguous", None)
    walk(r.root[1], "")
    return res
