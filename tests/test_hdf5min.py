"""av_aloha_amd/hdf5min.py: the episode files of record_sim_episodes.py:186-206 written and read back without h5py (it is not
installed in the build image), structures checked against the HDF5 file-format specification; where h5py exists, files cross
both ways (tests skipped otherwise)."""
import struct

import numpy as np
import pytest

from av_aloha_amd import harness, hdf5min


def episode(T=7, H=12, W=20, cams=("zed_cam", "cam_high")):
    rng = np.random.default_rng(5)
    d = {"/observations/qpos": rng.normal(size=(T, 21)).astype(np.float32), "/observations/qvel": rng.normal(size=(T, 21)).astype(np.float32),
         "/observations/all_qpos": rng.normal(size=(T, 37)).astype(np.float32), "/action": rng.normal(size=(T, 21)).astype(np.float32)}
    for c in cams:
        d[f"/observations/images/{c}"] = rng.integers(0, 256, size=(T, H, W, 3), dtype=np.uint8)
    return d


def test_episode_layout_written_and_read_back(tmp_path):
    data = episode()
    path = harness.save_episode(data, str(tmp_path), 4, use_h5py=False)
    assert path.endswith("episode_4.hdf5")
    back, attrs = hdf5min.read(path)
    assert set(back) == set(data) and list(attrs) == ["sim"] and attrs["sim"] is np.True_ or bool(attrs["sim"]) is True
    for k, v in data.items():
        assert back[k].dtype == v.dtype and np.array_equal(back[k], v), k
    lay = hdf5min.layout(path)
    assert lay["/observations/images/zed_cam"] == ("chunked", (1, 12, 20, 3))           # record_sim_episodes.py:197-198
    assert lay["/observations/qpos"] == ("contiguous", None) and lay["/action"] == ("contiguous", None)
    assert harness.load_episode(path).keys() == data.keys()


def test_file_structures_follow_the_format_specification(tmp_path):
    path = str(tmp_path / "f.hdf5")
    hdf5min.write(path, {"/a": np.arange(6, dtype=np.float32).reshape(2, 3), "/g/b": np.arange(4, dtype=np.uint8)}, attrs={"sim": np.bool_(True)})
    b = open(path, "rb").read()
    assert b[:8] == b"\x89HDF\r\n\x1a\n" and b[8] == 0                 # format signature, superblock version 0
    assert b[13] == 8 and b[14] == 8                                     # sizes of offsets and lengths
    leaf_k, int_k = struct.unpack_from("<HH", b, 16)
    assert (leaf_k, int_k) == (hdf5min.GROUP_LEAF_K, hdf5min.GROUP_INTERNAL_K)
    base, free, eof, drv = struct.unpack_from("<QQQQ", b, 24)
    assert base == 0 and free == hdf5min.UNDEF and drv == hdf5min.UNDEF and eof == len(b) and eof % 8 == 0
    name_off, root_oh, cache, _, btree, heap = struct.unpack_from("<QQIIQQ", b, 56)
    assert cache == 1 and b[btree:btree + 4] == b"TREE" and b[heap:heap + 4] == b"HEAP"
    # root object header: version 1, a symbol-table message pointing at the same B-tree / heap, one attribute message
    ver, _, nmsg, refs, size = struct.unpack_from("<BBHII", b, root_oh)
    assert (ver, nmsg, refs) == (1, 2, 1) and size % 8 == 0
    t0, s0 = struct.unpack_from("<HH", b, root_oh + 16)
    assert t0 == 0x0011 and struct.unpack_from("<QQ", b, root_oh + 24) == (btree, heap)
    t1, s1 = struct.unpack_from("<HH", b, root_oh + 24 + s0)
    assert t1 == 0x000C
    # local heap: free-list head 1 (= no free block), the names at 8-byte offsets, empty name at offset 0
    hsize, hfree, hseg = struct.unpack_from("<QQQ", b, heap + 8)
    assert hfree == 1 and hsize % 8 == 0 and b[hseg:hseg + 8] == b"\0" * 8 and b[hseg + 8:hseg + 10] == b"a\0" and b[hseg + 16:hseg + 18] == b"g\0"
    # group B-tree: node type 0, level 0, one child = a symbol-table node with two entries sorted by name
    ntype, level, used, left, right = struct.unpack_from("<BBHQQ", b, btree + 4)
    assert (ntype, level, used, left, right) == (0, 0, 1, hdf5min.UNDEF, hdf5min.UNDEF)
    key0, snod, key1 = struct.unpack_from("<QQQ", b, btree + 24)
    assert key0 == 0 and key1 == 16 and b[snod:snod + 4] == b"SNOD" and b[snod + 4] == 1 and struct.unpack_from("<H", b, snod + 6)[0] == 2
    e0 = struct.unpack_from("<QQII", b, snod + 8)
    e1 = struct.unpack_from("<QQII", b, snod + 48)
    assert e0[0] == 8 and e0[2] == 0 and e1[0] == 16 and e1[2] == 1                      # "a": a data set, "g": a group (cached B-tree / heap)
    # data set "a": dataspace v1 (rank 2, 2 x 3), IEEE float32 little endian, fill value v2, contiguous layout v3
    oh = e0[1]
    p, msgs = oh + 16, {}
    for _ in range(struct.unpack_from("<H", b, oh + 2)[0]):
        t, s = struct.unpack_from("<HH", b, p)
        msgs[t] = b[p + 8:p + 8 + s]
        p += 8 + s
    assert set(msgs) == {0x0001, 0x0003, 0x0005, 0x0008}
    assert msgs[1][:3] == bytes([1, 2, 0]) and struct.unpack_from("<QQ", msgs[1], 8) == (2, 3)
    assert msgs[3][0] == 0x11 and msgs[3][1] == 0x20 and msgs[3][2] == 31 and struct.unpack_from("<IHHBBBBI", msgs[3], 4) == (4, 0, 32, 23, 8, 0, 23, 127)
    assert msgs[8][0] == 3 and msgs[8][1] == 1
    addr, nbytes = struct.unpack_from("<QQ", msgs[8], 2)
    assert nbytes == 24 and np.array_equal(np.frombuffer(b, "<f4", 6, addr), np.arange(6, dtype=np.float32))


def test_chunk_tree_with_more_chunks_than_one_node_holds(tmp_path):
    """An episode has 301-401 time steps and one image chunk per step: the chunk B-tree (2 K = 64 children per node) gets a second level."""
    T = 150
    img = (np.arange(T * 2 * 5 * 3) % 251).astype(np.uint8).reshape(T, 2, 5, 3)
    path = str(tmp_path / "c.hdf5")
    hdf5min.write(path, {"/observations/images/cam": img}, chunks={"/observations/images/cam": (1, 2, 5, 3)})
    back, _ = hdf5min.read(path)
    assert np.array_equal(back["/observations/images/cam"], img)
    b = open(path, "rb").read()
    roots = [i for i in range(0, len(b) - 8, 8) if b[i:i + 4] == b"TREE" and b[i + 4] == 1]
    levels = sorted(b[i + 5] for i in roots)
    assert levels == [0, 0, 0, 1]                                       # 64 + 64 + 22 chunks under one level-1 node
    top = [i for i in roots if b[i + 5] == 1][0]
    assert struct.unpack_from("<H", b, top + 6)[0] == 3
    klen = 8 + 8 * 5
    keys = [struct.unpack_from("<IIQ", b, top + 24 + k * (klen + 8)) for k in range(4)]
    assert [k[2] for k in keys] == [0, 64, 128, 150] and keys[0][0] == 30 and keys[0][1] == 0
    # ragged first axis: chunks of 4 rows over 10 rows
    a = np.arange(10 * 3, dtype=np.float64).reshape(10, 3)
    hdf5min.write(path, {"/x": a}, chunks={"/x": (4, 3)})
    assert np.array_equal(hdf5min.read(path)[0]["/x"], a)


def test_dtypes_and_errors(tmp_path):
    path = str(tmp_path / "d.hdf5")
    data = {"/i32": np.arange(5, dtype=np.int32), "/f64": np.linspace(0, 1, 4), "/u8": np.arange(3, dtype=np.uint8), "/flag": np.array([True, False])}
    hdf5min.write(path, data, attrs={"n": np.int64(7), "x": np.float32(0.5)})
    back, attrs = hdf5min.read(path)
    for k, v in data.items():
        assert back[k].dtype == v.dtype and np.array_equal(back[k], v)
    assert attrs["n"] == 7 and attrs["x"] == np.float32(0.5)
    with pytest.raises(TypeError):
        hdf5min.write(path, {"/s": np.array(["a"])})
    (tmp_path / "bad").write_bytes(b"not hdf5")
    with pytest.raises(ValueError):
        hdf5min.read(str(tmp_path / "bad"))


def test_crosses_h5py_both_ways_where_it_is_installed(tmp_path):
    h5py = pytest.importorskip("h5py", reason="h5py is not installed in the build image; the byte-level test above checks the structures instead")
    data = episode()
    path = harness.save_episode(data, str(tmp_path), 0, use_h5py=False)
    with h5py.File(path, "r") as root:
        assert bool(root.attrs["sim"]) is True
        assert root["/observations/images/zed_cam"].chunks == (1, 12, 20, 3)
        for k, v in data.items():
            assert np.array_equal(root[k][()], v) and root[k].dtype == v.dtype
    path2 = harness.save_episode(data, str(tmp_path / "h"), 1, use_h5py=True)
    back, attrs = hdf5min.read(path2)
    assert bool(attrs["sim"]) is True
    for k, v in data.items():
        assert np.array_equal(back[k], v)


def _libhdf5():
    """The C library itself, where the image has one (no h5py needed): /opt/conda/lib/libhdf5.so* through ctypes."""
    import ctypes as C
    import glob
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            try:
                L = C.CDLL(p)
                if hasattr(L, "H5Fopen") and hasattr(L, "H5Dread"):
                    return L
            except OSError:
                continue
    return None


def _libhdf5_check(path, data):
    """Opens `path` with the real HDF5 library and compares every data set with `data`; False where no libhdf5 is installed."""
    import ctypes as C
    L = _libhdf5()
    if L is None:
        return False
    hid = C.c_int64
    L.H5open.restype = C.c_int
    L.H5Fopen.restype = hid; L.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    L.H5Dopen2.restype = hid; L.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    L.H5Dget_space.restype = hid; L.H5Dget_space.argtypes = [hid]
    L.H5Sget_simple_extent_ndims.restype = C.c_int; L.H5Sget_simple_extent_ndims.argtypes = [hid]
    L.H5Sget_simple_extent_dims.restype = C.c_int; L.H5Sget_simple_extent_dims.argtypes = [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.H5Dread.restype = C.c_int; L.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    L.H5Aexists.restype = C.c_int; L.H5Aexists.argtypes = [hid, C.c_char_p]
    for fn in ("H5Dclose", "H5Sclose", "H5Fclose"):
        getattr(L, fn).restype = C.c_int; getattr(L, fn).argtypes = [hid]
    assert L.H5open() >= 0
    native = {np.dtype(np.float32): hid.in_dll(L, "H5T_NATIVE_FLOAT_g").value, np.dtype(np.uint8): hid.in_dll(L, "H5T_NATIVE_UCHAR_g").value}
    f = L.H5Fopen(path.encode(), 0, 0)                                     # H5F_ACC_RDONLY, H5P_DEFAULT
    assert f >= 0, "libhdf5 does not accept the file"
    assert L.H5Aexists(f, b"sim") > 0
    for name, want in data.items():
        d = L.H5Dopen2(f, name.encode(), 0)
        assert d >= 0, name
        sp = L.H5Dget_space(d)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * nd)()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        assert tuple(dims) == want.shape, name
        got = np.empty(want.shape, dtype=want.dtype)
        assert L.H5Dread(d, native[want.dtype], 0, 0, 0, got.ctypes.data) >= 0, name      # H5S_ALL, H5S_ALL, H5P_DEFAULT
        assert np.array_equal(got, want), name
        L.H5Sclose(sp); L.H5Dclose(d)
    L.H5Fclose(f)
    return True


def test_real_libhdf5_reads_what_hdf5min_writes(tmp_path):
    """An episode file of the home-made writer opened by the real HDF5 library (H5Fopen / H5Dopen2 / H5Dget_space / H5Dread /
    H5Aexists): every dataset -- the chunked u8 image stacks and the contiguous f32 tables -- comes back identical and the root
    carries the attribute `sim` (record_sim_episodes.py:186-206).  Skipped where no libhdf5 is installed."""
    data = episode(T=5, H=24, W=32)
    path = harness.save_episode(data, str(tmp_path), 0, use_h5py=False)
    if not _libhdf5_check(path, data):
        pytest.skip("no libhdf5 shared library in this image")


def test_streamed_episode_equals_the_one_written_at_once(tmp_path):
    """hdf5min.StreamWriter: the image frames of an episode appended a step at a time, the cameras interleaved (what a recording that keeps no
    images in memory does), then the tables and the attribute: reads back as the same episode (hdf5min's reader; the real library where there
    is one); enough frames for a chunk B-tree of two levels; an aborted writer leaves no file."""
    from av_aloha_amd import hdf5min
    data = episode(T=150, H=6, W=10)
    imgs = {k: v for k, v in data.items() if "/images/" in k}
    rest = {k: v for k, v in data.items() if "/images/" not in k}
    p = str(tmp_path / "episode_0.hdf5")
    w = hdf5min.StreamWriter(p)
    for t in range(150):
        for k, v in imgs.items():
            w.append(k, v[t])
    w.finish(rest, attrs={"sim": np.bool_(True)})
    got, attrs = hdf5min.read(p)
    assert set(got) == set(data) and bool(attrs["sim"])
    for k, v in data.items():
        assert got[k].dtype == v.dtype and np.array_equal(got[k], v), k
    _libhdf5_check(p, data)
    with pytest.raises(AssertionError):
        w2 = hdf5min.StreamWriter(str(tmp_path / "x.hdf5"))
        w2.append("/a", np.zeros((2, 3), np.uint8))
        try:
            w2.append("/a", np.zeros((2, 4), np.uint8))
        finally:
            w2.abort()
    assert not (tmp_path / "x.hdf5").exists()


def test_a_failed_write_leaves_the_existing_file_alone(tmp_path):
    """write() goes through <path>.part and a rename: a data set name that passes through another data set makes finish() fail, the good
    file at the destination is untouched and no partial file stays behind."""
    import os
    from av_aloha_amd import hdf5min
    p = str(tmp_path / "episode_0.hdf5")
    hdf5min.write(p, {"/action": np.arange(6, dtype=np.float32).reshape(2, 3)}, {"sim": True})
    good = open(p, "rb").read()
    with pytest.raises(AssertionError):
        hdf5min.write(p, {"/a": np.zeros(3, dtype=np.float32), "/a/b": np.zeros(3, dtype=np.float32)})
    assert open(p, "rb").read() == good
    assert sorted(os.listdir(tmp_path)) == ["episode_0.hdf5"]
    np.testing.assert_array_equal(hdf5min.read(p)[0]["/action"], np.arange(6, dtype=np.float32).reshape(2, 3))
