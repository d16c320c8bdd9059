"""JPEG-2000 input rasters for the zonal loop.

The reference's live caller feeds BD ORTHO ``*.jp2`` mosaics (scripts/run_fast_aigle_segmentation.py:88;
flair_zonal_detection/inference.py:60 globs ``*.jp2`` first) and reads them through rasterio, i.e. GDAL's JP2OpenJPEG
driver (dataset.py:89-117 windowed boundless reads).  GDAL is not in the build image.  This module reads the same
files without it:

* the JP2 container (ISO/IEC 15444-1 Annex I box structure) and its georeferencing are parsed here: the GeoJP2 ``uuid``
  box (b14bf8bd-083d-4b43-a5ae-8cd7d5a6ce03, a degenerate GeoTIFF whose directory carries ModelPixelScale /
  ModelTiepoint / ModelTransformation and the GeoKey directory -- parsed by geotiff._parse_ifd), else an ESRI world
  file next to the raster (``.j2w`` / ``.jp2w`` / ``.wld`` / ``.jgw``, six lines), else rasterio's identity transform;
* the code-stream is decoded by OpenJPEG -- the decoder library GDAL's driver wraps -- through Pillow's Jpeg2K
  plugin.  Pillow exposes no region decode, but JPEG-2000 mosaics are TILED code-streams (BD ORTHO: 1024 or 2048 px
  tiles), and a tile is independently decodable: ``CodestreamIndex`` walks the SOT markers once (12 bytes per
  tile-part), and a window read assembles, per intersecting code-stream tile, a ONE-TILE code-stream -- SOC, the SIZ
  segment rewritten to the tile's extent at the origin, the rest of the main header verbatim, the tile's tile-parts
  with their tile index set to 0, EOC -- which Pillow decodes in a pool of worker PROCESSES (its decoder holds the GIL) into an
  LRU cache.  Reads are therefore lazy (the zonal loop touches each tile of the mosaic about once, and announces the
  footprint of its next batches through ``prefetch_bounds``) and parallel instead of one single-threaded 110 s decode
  of a 25 000 x 25 000 mosaic (round 2).  Relocating a tile to the origin keeps
  its wavelet / precinct / code-block partition only when the tile origin is a multiple of every partition size,
  which holds for power-of-two tile sizes on an un-shifted grid; anything else (untiled files, shifted grids, odd tile
  sizes, packed packet headers in the main header) falls back to the whole-image decode of round 2, once per raster;
* the duck-typed surface is ``RasterBase``'s (read / read_bounds / bounds / res / count / dtypes / profile / crs), the
  same as ``GeoTiffRaster``'s.

Raises ``Jp2Error`` with the reason when Pillow was built without OpenJPEG or the file is not JPEG-2000.
"""
from __future__ import annotations

import collections
import io
import os
import struct
import threading
from typing import Dict, Iterator, Optional, Tuple

import numpy as np

from flair_zonal_detection.geotiff import (_GEOKEYS, _PIXSCALE, _TIEPOINT, _TRANSFORM, Affine, _epsg_from_geokeys,
                                           _parse_ifd)
from flair_zonal_detection.raster import RasterBase

JP2_SIGNATURE = b"\x00\x00\x00\x0cjP  \r\n\x87\n"
J2K_SOC = b"\xff\x4f\xff\x51"  # raw code-stream: SOC + SIZ markers
GEOJP2_UUID = bytes.fromhex("b14bf8bd083d4b43a5ae8cd7d5a6ce03")
WORLD_FILE_SUFFIXES = (".j2w", ".jp2w", ".wld", ".jgw")


class Jp2Error(ValueError):
    pass


def is_jpeg2000(path: str) -> bool:
    with open(path, "rb") as f:
        head = f.read(12)
    return head == JP2_SIGNATURE or head[:4] == J2K_SOC


def openjpeg_available() -> bool:
    try:
        from PIL import features
        return bool(features.check_codec("jpg_2000"))
    except Exception:  # Pillow missing or too old to answer
        return False


def iter_boxes(f, start: int = 0, end: Optional[int] = None) -> Iterator[Tuple[bytes, int, int]]:
    """(type, payload offset, payload length) of the boxes between two file offsets (LBox = 1: 64-bit XLBox follows;
    LBox = 0: the box runs to the end of the file)"""
    if end is None:
        f.seek(0, os.SEEK_END)
        end = f.tell()
    pos = start
    while pos + 8 <= end:
        f.seek(pos)
        lbox, tbox = struct.unpack(">I4s", f.read(8))
        hdr = 8
        if lbox == 1:
            (lbox,) = struct.unpack(">Q", f.read(8))
            hdr = 16
        elif lbox == 0:
            lbox = end - pos
        if lbox < hdr or pos + lbox > end:
            raise Jp2Error(f"malformed box '{tbox.decode('latin-1')}' at offset {pos} (length {lbox})")
        yield tbox, pos + hdr, lbox - hdr
        pos += lbox


class CodestreamIndex:
    """Where each tile of a JPEG-2000 code-stream lives in the file (ISO/IEC 15444-1 Annex A marker syntax).

    Attributes: width / height / ncomp, tile size (xt, yt), grid (ntx, nty), ``main`` = the main-header marker segments
    between SIZ and the first SOT (COD, QCD, COC, QCC, RGN, POC, CRG, COM; TLM / PLM are dropped: they describe tile-part
    lengths of the whole stream), ``parts[t]`` = [(file offset of the SOT marker, tile-part length)], ``lazy`` = whether
    one-tile streams may be cut from it (see the module docstring) and, if not, ``why``."""

    def __init__(self, f, start: int, end: int):
        f.seek(start)
        if f.read(2) != b"\xff\x4f":
            raise Jp2Error("code-stream does not start with SOC")
        marker, lsiz = struct.unpack(">HH", f.read(4))
        if marker != 0xFF51:
            raise Jp2Error("SIZ does not follow SOC")
        siz = f.read(lsiz - 2)
        (self.rsiz, xs, ys, xo, yo, self.xt, self.yt, xto, yto, self.ncomp) = struct.unpack(">HIIIIIIIIH", siz[:36])
        self.comp_spec = siz[36:36 + 3 * self.ncomp]
        self.width, self.height = xs - xo, ys - yo
        self.ntx = -(-(xs - xto) // self.xt)
        self.nty = -(-(ys - yto) // self.yt)
        self.lazy, self.why = True, ""
        if (xo, yo, xto, yto) != (0, 0, 0, 0):
            self.lazy, self.why = False, "image or tile grid offset"
        elif self.ntx * self.nty == 1:
            self.lazy, self.why = False, "single tile"
        elif (self.xt & (self.xt - 1)) or (self.yt & (self.yt - 1)):
            self.lazy, self.why = False, f"tile size {self.xt} x {self.yt} is not a power of two"
        elif any(b != 1 for b in self.comp_spec[1::3] + self.comp_spec[2::3]):
            self.lazy, self.why = False, "sub-sampled components"
        main = bytearray()
        pos = f.tell()
        while True:
            f.seek(pos)
            hdr = f.read(4)
            if len(hdr) < 4:
                raise Jp2Error("code-stream ends inside the main header")
            marker, ln = struct.unpack(">HH", hdr)
            if marker == 0xFF90:  # SOT: the main header is over
                break
            if marker in (0xFF60,):  # PPM: packet headers of every tile live in the main header
                self.lazy, self.why = False, "packed packet headers (PPM)"
            if marker not in (0xFF55, 0xFF57, 0xFF60):  # TLM, PLM, PPM are not copied
                main += hdr + f.read(ln - 2)
            pos += 2 + ln
        self.main = bytes(main)
        self.parts: Dict[int, list] = {}
        while pos + 12 <= end:
            f.seek(pos)
            marker, lsot, isot, psot, tpsot, tnsot = struct.unpack(">HHHIBB", f.read(12))
            if marker == 0xFFD9:  # EOC
                break
            if marker != 0xFF90 or lsot != 10:
                raise Jp2Error(f"expected SOT at offset {pos}, found marker 0x{marker:04x}")
            if psot == 0:  # the last tile-part may run to EOC
                psot = end - pos - 2
            self.parts.setdefault(isot, []).append((pos, psot))
            pos += psot
            if not self.lazy:
                break  # nothing will be cut from this stream: no need to walk it

    def tile_stream(self, f, t: int) -> Tuple[bytes, int, int]:
        """(one-tile code-stream, tile width, tile height) of tile t = ty * ntx + tx"""
        tx, ty = t % self.ntx, t // self.ntx
        w = min(self.xt, self.width - tx * self.xt)
        h = min(self.yt, self.height - ty * self.yt)
        siz = struct.pack(">HIIIIIIIIH", self.rsiz, w, h, 0, 0, self.xt, self.yt, 0, 0, self.ncomp) + self.comp_spec
        out = bytearray(b"\xff\x4f" + struct.pack(">HH", 0xFF51, len(siz) + 2) + siz + self.main)
        parts = self.parts.get(t)
        if not parts:
            raise Jp2Error(f"tile {t} has no tile-part in the code-stream")
        for off, ln in parts:
            f.seek(off)
            buf = bytearray(f.read(ln))
            struct.pack_into(">H", buf, 4, 0)  # Isot: this is tile 0 of the cut stream
            out += buf
        out += b"\xff\xd9"
        return bytes(out), w, h


def _decode_stream(data: bytes) -> np.ndarray:
    """[H, W, C] pixels of a (one-tile) code-stream.  Module-level and argument-free of any raster state: it runs in
    the worker processes of ``_pool`` (Pillow's JPEG-2000 decoder holds the GIL, threads gave 1.0x; eight processes
    7.5x in the build container)."""
    from PIL import Image
    with Image.open(io.BytesIO(data)) as im:  # a tile is far below Pillow's pixel guard
        im.load()
        arr = np.asarray(im)
    return arr[:, :, None] if arr.ndim == 2 else arr


_POOL = None
_POOL_LOCK = threading.Lock()


def _pool():
    """Decoder processes, started on first use with the ``spawn`` method (a process that has initialised the GPU must
    not be forked) and shared by every raster of the process.  FFA_JP2_PROCS sets their number (default: the cores
    the process may use, at most 16); 0 decodes in the calling thread."""
    global _POOL
    n = os.environ.get("FFA_JP2_PROCS")
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    n = min(16, avail) if n is None else int(n)
    if n <= 0:
        return None
    with _POOL_LOCK:
        if _POOL is None:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            _POOL = ProcessPoolExecutor(max_workers=n, mp_context=mp.get_context("spawn"))
        return _POOL


def _georef_from_tiff(buf: bytes) -> Optional[Dict]:
    """left / top / xres / yres / epsg from the degenerate GeoTIFF of a GeoJP2 box"""
    if len(buf) < 8 or buf[:2] not in (b"II", b"MM"):
        return None
    bo = "<" if buf[:2] == b"II" else ">"
    (magic,) = struct.unpack_from(bo + "H", buf, 2)
    if magic != 42:
        return None
    (ifd,) = struct.unpack_from(bo + "I", buf, 4)
    try:
        t, _ = _parse_ifd(buf, bo, False, ifd)
    except (struct.error, IndexError):
        return None
    g: Dict = {"epsg": _epsg_from_geokeys(tuple(int(v) for v in t.get(_GEOKEYS, ())))}
    if _PIXSCALE in t and _TIEPOINT in t:
        sx, sy = float(t[_PIXSCALE][0]), float(t[_PIXSCALE][1])
        i, j, _, x, y, _ = (float(v) for v in t[_TIEPOINT][:6])
        g.update(left=x - i * sx, top=y + j * sy, xres=sx, yres=sy)
    elif _TRANSFORM in t:
        m = [float(v) for v in t[_TRANSFORM]]
        if m[1] != 0.0 or m[4] != 0.0 or m[5] >= 0.0:
            raise Jp2Error("rotated / south-up rasters are not supported")
        g.update(left=m[3], top=m[7], xres=m[0], yres=-m[5])
    else:
        return g if g["epsg"] else None
    return g


def _georef_from_world_file(path: str) -> Optional[Dict]:
    stem = os.path.splitext(path)[0]
    for suf in WORLD_FILE_SUFFIXES:
        for cand in (stem + suf, stem + suf.upper()):
            if os.path.exists(cand):
                vals = [float(v.replace(",", ".")) for v in open(cand).read().split()[:6]]
                if len(vals) != 6:
                    raise Jp2Error(f"{cand}: a world file holds six numbers")
                a, d, b, e, c, f = vals  # x = a col + b row + c, y = d col + e row + f, (c, f) = CENTRE of pixel (0, 0)
                if b != 0.0 or d != 0.0 or e >= 0.0:
                    raise Jp2Error(f"{cand}: rotated / south-up rasters are not supported")
                return {"left": c - a / 2, "top": f - e / 2, "xres": a, "yres": -e, "epsg": None}
    return None


class Jp2Raster(RasterBase):
    """Read-only JPEG-2000 raster (JP2 file or raw code-stream), decoded on first use."""

    def __init__(self, path: str, default_crs: Optional[str] = None):
        self.path = path
        self.closed = False
        self._data: Optional[np.ndarray] = None
        if not is_jpeg2000(path):
            raise Jp2Error(f"{path}: not a JPEG-2000 file")
        if not openjpeg_available():
            raise Jp2Error(f"{path}: JPEG-2000 needs Pillow built with OpenJPEG (or rasterio / GDAL); neither is "
                           "available in this environment")
        geo = None
        self.index: Optional[CodestreamIndex] = None
        self._tiles: "collections.OrderedDict[int, np.ndarray]" = collections.OrderedDict()
        self._lock = threading.Lock()
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(0)
            cs = None
            if f.read(12) == JP2_SIGNATURE:
                for tbox, off, n in iter_boxes(f):
                    if tbox == b"uuid" and n >= 16 and geo is None:
                        f.seek(off)
                        if f.read(16) == GEOJP2_UUID:
                            geo = _georef_from_tiff(f.read(n - 16))
                    elif tbox == b"jp2c":
                        cs = (off, off + n)
                        break  # georeferencing boxes precede the code-stream in every writer's layout; stop scanning
            else:
                cs = (0, size)
            if cs is not None:
                try:
                    self.index = CodestreamIndex(f, *cs)
                except (Jp2Error, struct.error):
                    self.index = None  # an unusual stream: Pillow / OpenJPEG still gets its chance on the whole file
        wf = _georef_from_world_file(path)
        if geo is None or "left" not in geo:  # the embedded box wins over a sidecar, as in GDAL's default order
            if wf is not None:
                wf["epsg"] = (geo or {}).get("epsg")
                geo = wf
        from PIL import Image
        # header only: size and mode, nothing is decoded here.  Pillow checks the pixel count against
        # MAX_IMAGE_PIXELS (~179 MP) inside Image.open and raises DecompressionBombError above twice that: a 25 000 x
        # 25 000 BD ORTHO mosaic is 625 MP, so the guard is lifted around the header read as it is around the decode
        keep = Image.MAX_IMAGE_PIXELS
        Image.MAX_IMAGE_PIXELS = None
        try:
            with Image.open(path) as im:
                self.width, self.height = im.size
                mode = im.mode
        finally:
            Image.MAX_IMAGE_PIXELS = keep
        try:
            self.count, self.dtype = {"L": (1, np.uint8), "LA": (2, np.uint8), "RGB": (3, np.uint8),
                                      "RGBA": (4, np.uint8), "I;16": (1, np.uint16), "I;16L": (1, np.uint16),
                                      "I;16B": (1, np.uint16), "I": (1, np.int32)}[mode]
        except KeyError:
            raise Jp2Error(f"{path}: unsupported sample layout (Pillow mode {mode})") from None
        self.dtype = np.dtype(self.dtype)
        if geo is not None and "left" in geo:
            self.left, self.top, self._xres, self._yres = geo["left"], geo["top"], geo["xres"], geo["yres"]
        else:
            self.left, self.top, self._xres, self._yres = 0.0, float(self.height), 1.0, 1.0  # rasterio's identity
        epsg = (geo or {}).get("epsg")
        self.crs = f"EPSG:{epsg}" if epsg else default_crs
        self.nodata = None

    # ---- rasterio-like attributes -------------------------------------------------------------------------
    @property
    def res(self) -> Tuple[float, float]:
        return (self._xres, self._yres)

    @property
    def transform(self) -> Affine:
        return Affine(self._xres, 0.0, self.left, 0.0, -self._yres, self.top)

    @property
    def dtypes(self) -> Tuple[str, ...]:
        return (str(self.dtype),) * self.count

    @property
    def profile(self) -> dict:
        return {"driver": "JP2OpenJPEG", "height": self.height, "width": self.width, "count": self.count,
                "dtype": str(self.dtype), "crs": self.crs, "transform": self.transform, "nodata": self.nodata}

    # ---- pixels -------------------------------------------------------------------------------------------
    def _decoded(self) -> np.ndarray:
        if self._data is None:
            if self.closed:
                raise Jp2Error(f"{self.path}: raster is closed")
            from PIL import Image
            limit = Image.MAX_IMAGE_PIXELS
            Image.MAX_IMAGE_PIXELS = None  # ortho mosaics are far beyond Pillow's decompression-bomb guard
            try:
                with Image.open(self.path) as im:
                    im.load()
                    arr = np.asarray(im)
            finally:
                Image.MAX_IMAGE_PIXELS = limit
            if arr.ndim == 2:
                arr = arr[:, :, None]
            if arr.shape != (self.height, self.width, self.count):
                raise Jp2Error(f"{self.path}: decoded {arr.shape}, header said {(self.height, self.width, self.count)}")
            # band-major like every other raster of the loop (one transpose per mosaic, then plain row slices)
            self._data = np.ascontiguousarray(arr.transpose(2, 0, 1)).astype(self.dtype, copy=False)
        return self._data

    # ---- lazy path: one code-stream tile at a time ------------------------------------------------------------
    TILE_CACHE_BYTES = 1 << 30  # decoded tiles kept (LRU); the zonal loop revisits a tile only across its margins

    @property
    def lazy(self) -> bool:
        return self.index is not None and self.index.lazy

    def _cut(self, t: int):
        with open(self.path, "rb") as f:
            return self.index.tile_stream(f, t)

    def _finish_tile(self, t: int, arr: np.ndarray, w: int, h: int) -> np.ndarray:
        if arr.shape != (h, w, self.count):
            raise Jp2Error(f"{self.path}: tile {t} decoded to {arr.shape}, expected {(h, w, self.count)}")
        return np.ascontiguousarray(arr.transpose(2, 0, 1)).astype(self.dtype, copy=False)

    def _tiles_for(self, ids) -> Dict[int, np.ndarray]:
        with self._lock:
            have = {t: self._tiles[t] for t in ids if t in self._tiles}
            for t in have:
                self._tiles.move_to_end(t)
        missing = [t for t in ids if t not in have]
        if missing:
            cuts = [self._cut(t) for t in missing]
            pool = _pool() if len(missing) > 1 else None
            if pool is None:
                raw = [_decode_stream(c[0]) for c in cuts]
            else:
                raw = list(pool.map(_decode_stream, [c[0] for c in cuts]))
            fresh = [self._finish_tile(t, a, c[1], c[2]) for t, a, c in zip(missing, raw, cuts)]
            with self._lock:
                for t, a in zip(missing, fresh):
                    have[t] = self._tiles[t] = a
                total = sum(a.nbytes for a in self._tiles.values())
                while total > self.TILE_CACHE_BYTES and len(self._tiles) > len(ids):
                    _, old = self._tiles.popitem(last=False)
                    total -= old.nbytes
        return have

    def _block(self, bands, ys: int, ye: int, xs: int, xe: int, out: np.ndarray) -> None:
        if not self.lazy:
            data = self._decoded()
            for k, bnd in enumerate(bands):
                out[k] = data[bnd, ys:ye, xs:xe]
            return
        if self.closed:
            raise Jp2Error(f"{self.path}: raster is closed")
        ix = self.index
        tys = range(ys // ix.yt, (ye - 1) // ix.yt + 1)
        txs = range(xs // ix.xt, (xe - 1) // ix.xt + 1)
        tiles = self._tiles_for([ty * ix.ntx + tx for ty in tys for tx in txs])
        for ty in tys:
            y0, y1 = max(ys, ty * ix.yt), min(ye, (ty + 1) * ix.yt)
            for tx in txs:
                x0, x1 = max(xs, tx * ix.xt), min(xe, (tx + 1) * ix.xt)
                a = tiles[ty * ix.ntx + tx]
                for k, bnd in enumerate(bands):
                    out[k, y0 - ys:y1 - ys, x0 - xs:x1 - xs] = a[bnd, y0 - ty * ix.yt:y1 - ty * ix.yt,
                                                                 x0 - tx * ix.xt:x1 - tx * ix.xt]

    def prefetch_bounds(self, bounds) -> None:
        """decode, in parallel, the code-stream tiles under a geographic box that the loop is about to read piecewise"""
        if not self.lazy or self.closed:
            return
        l, b, r, t = bounds
        ix = self.index
        xs, xe = int((l - self.left) // self._xres), int(-(-(r - self.left) // self._xres))
        ys, ye = int((self.top - t) // self._yres), int(-(-(self.top - b) // self._yres))
        xs, ys, xe, ye = max(xs, 0), max(ys, 0), min(xe, self.width), min(ye, self.height)
        if xe > xs and ye > ys:
            self._tiles_for([ty * ix.ntx + tx for ty in range(ys // ix.yt, (ye - 1) // ix.yt + 1)
                             for tx in range(xs // ix.xt, (xe - 1) // ix.xt + 1)])

    def prefetch_rows(self, ys: int, ye: int) -> None:
        """decode every code-stream tile that rows [ys, ye) touch, in parallel (a reader thread calls this one tile row
        ahead of the zonal loop)"""
        if self.lazy and ye > ys:
            ix = self.index
            self._tiles_for([ty * ix.ntx + tx for ty in range(max(ys, 0) // ix.yt, min(ye - 1, self.height - 1) // ix.yt + 1)
                             for tx in range(ix.ntx)])

    def close(self) -> None:
        self._data = None
        self._tiles.clear()
        self.closed = True
