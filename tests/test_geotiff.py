"""GeoTIFF side of the zonal loop, CPU only: the libtiff-compatible LZW codec and
predictor of libflairhip (ffa_tiff_*), the reader against files written by an independent TIFF implementation
(Pillow / libtiff, when installed) and against hand-assembled big-endian / BigTIFF / strip files, and the writer's
window writes read back -- by our reader and by Pillow."""
import io
import struct

import numpy as np
import pytest

from flair_zonal_detection.geotiff import GeoTiffError, GeoTiffRaster, GeoTiffWriter
from flair_zonal_detection.raster import ArrayRaster, make_window, open_raster


def _lib():
    from flairhip import lib
    return lib.load()


def lzw_encode(a: np.ndarray) -> np.ndarray:
    L = _lib()
    a = np.ascontiguousarray(a, np.uint8)
    cap = L.ffa_tiff_lzw_bound(a.size)
    out = np.empty(cap, np.uint8)
    n = L.ffa_tiff_lzw_encode(a.ctypes.data, a.size, out.ctypes.data, cap)
    assert n > 0, n
    return out[:n].copy()


def lzw_decode(e: np.ndarray, size: int) -> np.ndarray:
    L = _lib()
    out = np.empty(max(size, 1), np.uint8)
    n = L.ffa_tiff_lzw_decode(e.ctypes.data, e.size, out.ctypes.data, size)
    assert n >= 0, n
    return out[:n]


def _cases():
    g = np.random.default_rng(0)
    return {
        "empty": np.zeros(0, np.uint8),
        "one": np.array([7], np.uint8),
        "zeros": np.zeros(65536, np.uint8),  # the 'KwKwK' case (code == next free code) all the way
        "noise": g.integers(0, 256, 70000, dtype=np.uint8),  # fills the 4094-entry table several times
        "classes": np.repeat(g.integers(0, 19, 3000, dtype=np.uint8), g.integers(1, 60, 3000)),
        "ramp": (np.arange(100000) % 251).astype(np.uint8),
    }


@pytest.mark.parametrize("name", list(_cases()))
def test_lzw_round_trip(name):
    a = _cases()[name]
    e = lzw_encode(a)
    assert e.size <= _lib().ffa_tiff_lzw_bound(a.size)
    assert np.array_equal(lzw_decode(e, a.size), a)
    if a.size > 8:  # a short destination stops the decoder without overrunning it
        assert np.array_equal(lzw_decode(e, a.size // 2), a[:a.size // 2])


def test_lzw_errors_are_reported_not_crashes():
    L = _lib()
    out = np.empty(64, np.uint8)
    bad = np.array([0x80, 0x3F, 0xFF, 0xFF, 0xFF], np.uint8)  # clear, then a code far past the table
    assert L.ffa_tiff_lzw_decode(bad.ctypes.data, bad.size, out.ctypes.data, 64) < 0
    assert b"lzw_decode" in L.ffa_last_error()
    a = np.arange(256, dtype=np.uint8)
    assert L.ffa_tiff_lzw_encode(a.ctypes.data, a.size, out.ctypes.data, 8) == -3  # FFA_ERR_WORKSPACE


@pytest.mark.parametrize("nbytes,stride", [(1, 1), (1, 3), (2, 1), (2, 4), (4, 2)])
def test_horizontal_predictor_round_trip(nbytes, stride):
    g = np.random.default_rng(nbytes * 10 + stride)
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[nbytes]
    a = g.integers(0, np.iinfo(dt).max, (5, 12 * stride), dtype=dt)
    b = a.copy()
    assert _lib().ffa_tiff_hpredict(b.ctypes.data, 5, b.shape[1], nbytes, stride, 0) == 0
    ref = a.copy()
    ref[:, stride:] = a[:, stride:] - a[:, :-stride]  # wraps modulo 2^bits like the C code
    assert np.array_equal(b, ref)
    assert _lib().ffa_tiff_hpredict(b.ctypes.data, 5, b.shape[1], nbytes, stride, 1) == 0
    assert np.array_equal(b, a)


# ---- reader against Pillow (libtiff) -----------------------------------------------------------------------------

PIL = pytest.importorskip("PIL.Image", reason="Pillow is the independent TIFF implementation of these tests")


def _pil_save(arr, path, **kw):
    from PIL import Image
    Image.fromarray(arr).save(path, format="TIFF", **kw)


@pytest.mark.parametrize("compression", [None, "tiff_lzw", "tiff_adobe_deflate"])
def test_reads_pillow_files(tmp_path, compression):
    g = np.random.default_rng(1)
    rgb = np.repeat(g.integers(0, 255, (150, 35, 3), dtype=np.uint8), 8, axis=1)  # 150 x 280, compressible
    p = str(tmp_path / "rgb.tif")
    _pil_save(rgb, p, **({"compression": compression} if compression else {}))
    with GeoTiffRaster(p) as r:
        assert (r.count, r.height, r.width) == (3, 150, 280) and r.dtypes == ("uint8",) * 3
        assert np.array_equal(r.read(), rgb.transpose(2, 0, 1))
        assert np.array_equal(r.read(2, window=make_window(10, 20, 50, 60)), rgb[20:80, 10:60, 1])
        assert r.crs is None and r.res == (1.0, 1.0)
    u16 = g.integers(0, 65535, (70, 90), dtype=np.uint16)
    p = str(tmp_path / "u16.tif")
    _pil_save(u16, p, **({"compression": compression} if compression else {}))
    with GeoTiffRaster(p) as r:
        assert r.dtypes == ("uint16",) and np.array_equal(r.read(1), u16)
    f32 = g.standard_normal((33, 47)).astype(np.float32)
    p = str(tmp_path / "f32.tif")
    _pil_save(f32, p, **({"compression": compression} if compression else {}))
    with GeoTiffRaster(p) as r:
        assert r.dtypes == ("float32",) and np.array_equal(r.read(1), f32)


def test_reads_pillow_lzw_with_predictor(tmp_path):
    from PIL import Image, TiffImagePlugin
    g = np.random.default_rng(2)
    img = np.cumsum(g.integers(0, 3, (64, 300, 3)), axis=1).astype(np.uint8)
    p = str(tmp_path / "pred.tif")
    ifd = TiffImagePlugin.ImageFileDirectory_v2()
    ifd[317] = 2
    Image.fromarray(img).save(p, format="TIFF", compression="tiff_lzw", tiffinfo=ifd)
    with GeoTiffRaster(p) as r:
        assert r.tags[317][0] == 2
        assert np.array_equal(r.read(), img.transpose(2, 0, 1))


# ---- reader against hand-assembled files (byte order, BigTIFF, strips, band-separate planes) -----------------------

def _assemble(bo, big, arr, planar, rows_per_strip, geo=None):
    """uncompressed strip TIFF of arr [count, H, W] (uint16) written field by field"""
    count, H, W = arr.shape
    dt = np.dtype(bo + "u2")
    strips = []
    if planar == 1:
        pix = np.ascontiguousarray(arr.transpose(1, 2, 0)).astype(dt)
        for y in range(0, H, rows_per_strip):
            strips.append(pix[y:y + rows_per_strip].tobytes())
    else:
        for b in range(count):
            for y in range(0, H, rows_per_strip):
                strips.append(arr[b, y:y + rows_per_strip].astype(dt).tobytes())
    o = "Q" if big else "I"
    head = (b"II" if bo == "<" else b"MM") + (struct.pack(bo + "HHHQ", 43, 8, 0, 0) if big else struct.pack(bo + "HI", 42, 0))
    data, offs, pos = b"", [], len(head)
    for s in strips:
        offs.append(pos + len(data))
        data += s
    LONGT = 16 if big else 4
    ent = [(256, 3, (W,)), (257, 3, (H,)), (258, 3, (16,) * count), (259, 3, (1,)), (262, 3, (1,)),
           (273, LONGT, tuple(offs)), (277, 3, (count,)), (278, 3, (rows_per_strip,)),
           (279, LONGT, tuple(len(s) for s in strips)), (284, 3, (planar,)), (339, 3, (1,) * count)]
    if geo:
        ent += [(33550, 12, (geo["res"], geo["res"], 0.0)), (33922, 12, (0.0, 0.0, 0.0, geo["left"], geo["top"], 0.0)),
                (34735, 3, (1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, geo["epsg"]))]
    ent.sort()
    fmt = {3: "H", 4: "I", 12: "d", 16: "Q"}
    inl, esz = (8, 20) if big else (4, 12)
    ifd_off = len(head) + len(data)
    table = (8 if big else 2) + len(ent) * esz + (8 if big else 4)
    body, extra = b"", b""
    for tag, typ, vals in ent:
        payload = struct.pack(bo + fmt[typ] * len(vals), *vals)
        if len(payload) <= inl:
            field = payload.ljust(inl, b"\0")
        else:
            field = struct.pack(bo + o, ifd_off + table + len(extra))
            extra += payload
        body += struct.pack(bo + "HH" + o, tag, typ, len(vals)) + field
    blob = head + data + struct.pack(bo + ("Q" if big else "H"), len(ent)) + body + struct.pack(bo + o, 0) + extra
    blob = bytearray(blob)
    struct.pack_into(bo + o, blob, 8 if big else 4, ifd_off)
    return bytes(blob)


@pytest.mark.parametrize("bo", ["<", ">"])
@pytest.mark.parametrize("big", [False, True])
@pytest.mark.parametrize("planar", [1, 2])
def test_reads_strip_files_of_either_byte_order_and_offset_size(tmp_path, bo, big, planar):
    g = np.random.default_rng(5)
    arr = g.integers(0, 65535, (3, 37, 53)).astype(np.uint16)
    geo = {"left": 651992.4, "top": 6860417.8, "res": 0.2, "epsg": 2154}
    p = tmp_path / "t.tif"
    p.write_bytes(_assemble(bo, big, arr, planar, rows_per_strip=8, geo=geo))
    with open_raster(str(p)) as r:  # no rasterio in the image: open_raster picks the built-in reader
        assert isinstance(r, GeoTiffRaster)
        assert np.array_equal(r.read(), arr)
        assert np.array_equal(r.read([3, 1], window=make_window(5, 30, 20, 7)), arr[[2, 0], 30:37, 5:25])
        assert r.crs == "EPSG:2154" and r.res == (0.2, 0.2)
        b = r.bounds
        assert (b.left, b.top) == (651992.4, 6860417.8) and b.right == 651992.4 + 53 * 0.2
        # boundless window hanging over the top-left corner
        w = r.read([1], window=make_window(-4, -3, 10, 10), boundless=True, fill_value=9)
        assert (w[0, :3] == 9).all() and (w[0, :, :4] == 9).all() and np.array_equal(w[0, 3:, 4:], arr[0, :7, :6])


def test_read_bounds_matches_the_in_memory_raster(tmp_path):
    g = np.random.default_rng(6)
    arr = g.integers(0, 255, (4, 200, 260)).astype(np.uint8)
    left, top, res = 1000.0, 5000.0, 0.5
    p = str(tmp_path / "m.tif")
    with GeoTiffWriter(p, 260, 200, 4, left, top, res, crs="EPSG:2154") as w:
        for b in range(4):
            w.write(arr[b], b + 1)
    mem = ArrayRaster(arr, left, top, res)
    with GeoTiffRaster(p, cache_bytes=1 << 16) as r:  # tiny cache: blocks get evicted and decoded again
        for box, size in [((1010.0, 4950.0, 1042.0, 4982.0), 64),     # aligned, inside
                          ((990.0, 4980.0, 1022.0, 5012.0), 64),      # aligned, hangs over the top-left corner
                          ((1100.0, 4880.0, 1150.0, 4930.0), 64),     # 100 px box -> 64: bilinear
                          ((1120.0, 4890.0, 1140.0, 4910.0), 64),     # 40 px box -> 64: bilinear, upsampling
                          ((2000.0, 2000.0, 2032.0, 2032.0), 64)]:    # outside the raster
            assert np.array_equal(r.read_bounds([1, 3, 4], box, size), mem.read_bounds([1, 3, 4], box, size)), box


# ---- writer ---------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("compress", ["lzw", "deflate", None])
def test_window_writes_come_back_from_the_file(tmp_path, compress):
    g = np.random.default_rng(7)
    H, W = 300, 410  # not multiples of the 256-pixel block: partial edge blocks
    want = np.zeros((1, H, W), np.uint8)
    p = str(tmp_path / "pred.tif")
    ref = ArrayRaster(np.zeros((3, H, W), np.uint8), 651992.36, 6860417.84, 0.2)
    w = GeoTiffWriter.like(p, ref, 1, compress=compress)
    for _ in range(12):
        c, r = int(g.integers(0, W - 96)), int(g.integers(0, H - 96))
        tile = np.repeat(np.repeat(g.integers(0, 19, (12, 12), dtype=np.uint8), 8, 0), 8, 1)
        w.write(tile, 1, window=make_window(c, r, 96, 96))
        want[0, r:r + 96, c:c + 96] = tile
    w.close()
    w.close()  # idempotent
    with GeoTiffRaster(p) as r:
        assert np.array_equal(r.read(), want)
        assert r.crs == "EPSG:2154" and r.res == (0.2, 0.2) and tuple(r.bounds) == tuple(ref.bounds)
        assert r.profile["tiled"] and r.profile["blockxsize"] == 256 and r.profile["compress"] == compress
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(p)), want[0])  # libtiff agrees on every byte


def test_multi_band_class_probabilities_and_scratch_file(tmp_path):
    g = np.random.default_rng(8)
    arr = g.integers(0, 255, (19, 70, 300)).astype(np.uint8)
    p = str(tmp_path / "prob.tif")
    w = GeoTiffWriter(p, 300, 70, 19, 10.0, 20.0, (0.5, 0.25), crs="EPSG:4326", scratch_above=1000)  # forces the memmap
    assert w._scratch is not None
    for b in range(19):
        w.write(arr[b], b + 1)
    scratch = w._scratch
    w.close()
    import os
    assert not os.path.exists(scratch) and not os.path.exists(p + ".part")
    with GeoTiffRaster(p) as r:
        assert r.count == 19 and r.res == (0.5, 0.25) and r.crs == "EPSG:4326" and r.profile["interleave"] == "band"
        assert np.array_equal(r.read(), arr)


def test_unsupported_files_raise(tmp_path):
    p = tmp_path / "x.jp2"
    p.write_bytes(b"\0\0\0\x0cjP  \r\n\x87\n" + b"\0" * 64)
    with pytest.raises(GeoTiffError, match="not a TIFF"):
        GeoTiffRaster(str(p))
    g = np.random.default_rng(9)
    q = str(tmp_path / "pb.tif")
    _pil_save(g.integers(0, 255, (20, 20), dtype=np.uint8), q, compression="packbits")
    with pytest.raises(GeoTiffError, match="PackBits"):
        GeoTiffRaster(q)
    with pytest.raises(GeoTiffError, match="compress"):
        GeoTiffWriter(str(tmp_path / "y.tif"), 4, 4, 1, 0.0, 4.0, 1.0, compress="jpeg")


def test_merge_shard_files_applies_parts_in_rank_order(tmp_path):
    from flair_zonal_detection.geotiff import WRITTEN_SUFFIX, merge_shard_files
    g = np.random.default_rng(10)
    H, W = 90, 300
    ref = ArrayRaster(np.zeros((1, H, W), np.uint8), 0.0, 90.0, 1.0)
    want = np.zeros((2, H, W), np.uint8)
    paths = []
    for rank, (c0, c1) in enumerate([(0, 120), (100, 220), (200, 300)]):  # overlapping column ranges
        p = str(tmp_path / f"part{rank}.tif")
        w = GeoTiffWriter.like(p, ref, 2)
        w.track_writes()
        for b in range(2):
            blk = g.integers(1, 200, (H - 10 * rank, c1 - c0), dtype=np.uint8)
            w.write(blk, b + 1, window=make_window(c0, 0, c1 - c0, H - 10 * rank))
            want[b, :H - 10 * rank, c0:c1] = blk
        w.close()
        paths.append(p)
        with GeoTiffRaster(p + WRITTEN_SUFFIX) as m:
            assert m.read(1).sum() == (H - 10 * rank) * (c1 - c0)
    out = merge_shard_files(paths, str(tmp_path / "merged.tif"), rows_per_pass=32)
    with GeoTiffRaster(out) as r:
        assert np.array_equal(r.read(), want)


def test_truncated_directory_is_a_geotiff_error(tmp_path):
    g = np.random.default_rng(12)
    blob = _assemble("<", False, g.integers(0, 9, (1, 8, 8)).astype(np.uint16), 1, 8)
    (ifd,) = struct.unpack_from("<I", blob, 4)
    p = tmp_path / "cut.tif"
    p.write_bytes(blob[:ifd + 20])  # directory cut short
    with pytest.raises(GeoTiffError, match="malformed"):
        GeoTiffRaster(str(p))
    q = tmp_path / "nodims.tif"
    q.write_bytes(b"II" + struct.pack("<HI", 42, 8) + struct.pack("<H", 0) + struct.pack("<I", 0))  # empty directory
    with pytest.raises(GeoTiffError, match="malformed"):
        GeoTiffRaster(str(q))
