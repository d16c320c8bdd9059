"""GeoTIFF reader / writer for the zonal loop (the raster I/O either side of the path).

The reference opens its rasters with rasterio (GDAL): windowed boundless reads of the input mosaics
(flair_zonal_detection/dataset.py:89-117) and LZW-compressed uint8 window writes of the prediction rasters
(inference.py:157-208 profile, :342-352 ``dst.write(..., window=Window(...))``).  GDAL is not in the build image, so
this module speaks the subset of TIFF 6.0 / BigTIFF / GeoTIFF 1.1 those files use and presents the same duck-typed
surface as ``raster.ArrayRaster`` (which a real rasterio dataset also satisfies):

* ``GeoTiffRaster(path)``: classic or BigTIFF, either byte order, strips or tiles, pixel- or band-interleaved,
  8/16/32/64-bit integer and float samples, compression none / LZW / Deflate, Predictor 1 or 2, north-up
  georeferencing (ModelPixelScale + ModelTiepoint, or an axis-aligned ModelTransformation), EPSG code from the
  GeoKey directory.  The file is memory-mapped; only the blocks a window touches are decoded (LRU cache), so a
  25 000 x 25 000 px mosaic is never resident as a whole.  JPEG-2000 / JPEG-in-TIFF need a codec GDAL would bring and
  raise a clear error.
* ``GeoTiffWriter``: the prediction raster.  Window writes land in an in-memory (or memory-mapped scratch) array;
  ``close()`` encodes 256 x 256 blocks with the LZW codec of libflairhip on a thread pool and writes the file once
  (GDAL's own window writes re-append rewritten blocks; writing once keeps the file dense).

The byte codecs (LZW, horizontal predictor) are C++ in csrc/tiff_codec.cpp behind ``ffa_tiff_*``.
"""
from __future__ import annotations

import mmap
import os
import struct
import tempfile
import zlib
from collections import OrderedDict, namedtuple
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from flair_zonal_detection.raster import ArrayRaster, BoundingBox, RasterBase

Affine = namedtuple("Affine", ["a", "b", "c", "d", "e", "f"])  # x = a*col + b*row + c ; y = d*col + e*row + f

# tag ids
_W, _H, _BITS, _COMP, _PHOTO, _STRIP_OFF, _SPP, _RPS, _STRIP_CNT = 256, 257, 258, 259, 262, 273, 277, 278, 279
_PLANAR, _PREDICTOR, _TW, _TL, _TILE_OFF, _TILE_CNT, _EXTRA, _FORMAT = 284, 317, 322, 323, 324, 325, 338, 339
_PIXSCALE, _TIEPOINT, _TRANSFORM, _GEOKEYS, _GEODOUBLES, _GEOASCII, _NODATA = 33550, 33922, 34264, 34735, 34736, 34737, 42113

_TYPE = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("I", 4), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2),
         9: ("i", 4), 10: ("i", 4), 11: ("f", 4), 12: ("d", 8), 16: ("Q", 8), 17: ("q", 8), 18: ("Q", 8)}


class GeoTiffError(ValueError):
    pass


def _codec():
    from flairhip import lib
    return lib.load()


_POOL: Optional[ThreadPoolExecutor] = None
_POOL_WORKERS = max(1, min(16, os.cpu_count() or 1))


def _pool() -> ThreadPoolExecutor:
    """shared decode / encode workers (ctypes calls and zlib release the GIL)"""
    global _POOL
    if _POOL is None:
        _POOL = ThreadPoolExecutor(_POOL_WORKERS, thread_name_prefix="geotiff")
    return _POOL


def _parse_ifd(buf, bo: str, big: bool, off: int) -> Tuple[Dict[int, tuple], int]:
    """tag -> tuple of values of the IFD at ``off``; also the offset of the next IFD (0 = none)"""
    if big:
        (n,) = struct.unpack_from(bo + "Q", buf, off)
        pos, esz, inl, cfmt = off + 8, 20, 8, "Q"
    else:
        (n,) = struct.unpack_from(bo + "H", buf, off)
        pos, esz, inl, cfmt = off + 2, 12, 4, "I"
    tags: Dict[int, tuple] = {}
    for k in range(n):
        e = pos + k * esz
        tag, typ = struct.unpack_from(bo + "HH", buf, e)
        (cnt,) = struct.unpack_from(bo + cfmt, buf, e + 4)
        if typ not in _TYPE:
            continue
        ch, sz = _TYPE[typ]
        per = 2 if typ in (5, 10) else 1
        nbytes = cnt * sz * per
        voff = e + 4 + struct.calcsize(cfmt)
        if nbytes > inl:
            (voff,) = struct.unpack_from(bo + cfmt, buf, voff)
        if typ == 2:
            tags[tag] = (bytes(buf[voff:voff + cnt]).split(b"\0")[0].decode("latin-1"),)
        elif typ == 7:
            tags[tag] = (bytes(buf[voff:voff + cnt]),)
        else:
            vals = struct.unpack_from(bo + ch * (cnt * per), buf, voff)
            if per == 2:
                vals = tuple(vals[i] / vals[i + 1] if vals[i + 1] else 0.0 for i in range(0, len(vals), 2))
            tags[tag] = vals
    (nxt,) = struct.unpack_from(bo + cfmt, buf, pos + n * esz)
    return tags, nxt


def _epsg_from_geokeys(keys: Sequence[int]) -> Optional[int]:
    """ProjectedCSTypeGeoKey (3072) if present and not user-defined, else GeographicTypeGeoKey (2048)"""
    if not keys or len(keys) < 4:
        return None
    found = {}
    for i in range(int(keys[3])):
        k = keys[4 + 4 * i: 8 + 4 * i]
        if len(k) == 4 and k[1] == 0 and k[2] == 1:
            found[k[0]] = k[3]
    for key in (3072, 2048):
        if found.get(key, 32767) not in (0, 32767):
            return int(found[key])
    return None


def _geokeys_for_epsg(epsg: int) -> Tuple[int, ...]:
    # EPSG geographic 2D systems live in 4000..4999 (4326 WGS 84, 4171 RGF93, 4258 ETRS89); the rest is projected
    geographic = 4000 <= epsg <= 4999
    return (1, 1, 0, 3, 1024, 0, 1, 2 if geographic else 1, 1025, 0, 1, 1, 2048 if geographic else 3072, 0, 1, epsg)


class GeoTiffRaster(RasterBase):
    """Read-only, memory-mapped GeoTIFF (first image of the file; overviews are ignored)."""

    def __init__(self, path: str, cache_bytes: int = 512 << 20):
        self.path = path
        self._cache: "OrderedDict[int, np.ndarray]" = OrderedDict()
        self._cache_bytes, self._cache_cap = 0, int(cache_bytes)
        self._f = open(path, "rb")
        try:
            self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError as e:
            self._f.close()
            raise GeoTiffError(f"{path}: empty file") from e
        mm = self._mm
        sig = bytes(mm[:4])
        if sig[:2] == b"II":
            bo = "<"
        elif sig[:2] == b"MM":
            bo = ">"
        else:
            self.close()
            raise GeoTiffError(f"{path}: not a TIFF file (JPEG-2000 and other GDAL formats need rasterio)")
        (magic,) = struct.unpack_from(bo + "H", mm, 2)
        if magic == 42:
            big = False
            (ifd,) = struct.unpack_from(bo + "I", mm, 4)
        elif magic == 43:
            big = True
            (ifd,) = struct.unpack_from(bo + "Q", mm, 8)
        else:
            self.close()
            raise GeoTiffError(f"{path}: bad TIFF magic {magic}")
        self._bo = bo
        try:
            t, _ = _parse_ifd(mm, bo, big, ifd)
            self._init_from_tags(t)
        except (struct.error, IndexError, KeyError) as e:  # truncated / inconsistent directory
            self.close()
            raise GeoTiffError(f"{path}: malformed TIFF directory ({type(e).__name__}: {e})") from e
        except GeoTiffError:
            self.close()
            raise

    def _init_from_tags(self, t: Dict[int, tuple]) -> None:
        path, bo = self.path, self._bo
        self.tags = t
        self.width, self.height = int(t[_W][0]), int(t[_H][0])
        self.count = int(t.get(_SPP, (1,))[0])
        bits = t.get(_BITS, (1,))
        fmt = t.get(_FORMAT, (1,))
        if len(set(bits)) != 1 or len(set(fmt)) != 1:
            raise GeoTiffError(f"{path}: bands of different sample types are not supported")
        kind = {1: "u", 2: "i", 3: "f"}.get(int(fmt[0]))
        if kind is None or int(bits[0]) not in (8, 16, 32, 64) or (kind == "f" and bits[0] < 32):
            raise GeoTiffError(f"{path}: unsupported sample type (format {fmt[0]}, {bits[0]} bits)")
        self.dtype = np.dtype(f"{bo}{kind}{int(bits[0]) // 8}")
        self._native = self.dtype.newbyteorder("=")
        self._comp = int(t.get(_COMP, (1,))[0])
        if self._comp not in (1, 5, 8, 32946):
            name = {6: "old JPEG", 7: "JPEG", 32773: "PackBits", 34712: "JPEG-2000", 50000: "ZSTD", 50001: "WebP",
                    34925: "LZMA"}.get(self._comp, str(self._comp))
            raise GeoTiffError(f"{path}: compression {name} is not supported (none / LZW / Deflate are)")
        self._pred = int(t.get(_PREDICTOR, (1,))[0])
        if self._pred not in (1, 2) or (self._pred == 2 and (kind == "f" or bits[0] == 64)):
            raise GeoTiffError(f"{path}: predictor {self._pred} is not supported for this sample type")
        self._planar = int(t.get(_PLANAR, (1,))[0]) if self.count > 1 else 1
        if _TW in t:
            self._bw, self._bh = int(t[_TW][0]), int(t[_TL][0])
            self._offs, self._cnts = t[_TILE_OFF], t[_TILE_CNT]
        else:
            self._bw = self.width
            self._bh = min(int(t.get(_RPS, (self.height,))[0]), self.height)
            self._offs, self._cnts = t[_STRIP_OFF], t[_STRIP_CNT]
        self._nbx = -(-self.width // self._bw)
        self._nby = -(-self.height // self._bh)
        want = self._nbx * self._nby * (self.count if self._planar == 2 else 1)
        if len(self._offs) != want or len(self._cnts) != want:
            raise GeoTiffError(f"{path}: {len(self._offs)} blocks listed, {want} expected")
        # georeferencing: north-up only (what the tile grid of slicing.py assumes)
        if _PIXSCALE in t and _TIEPOINT in t:
            sx, sy = float(t[_PIXSCALE][0]), float(t[_PIXSCALE][1])
            i, j, _, x, y, _ = (float(v) for v in t[_TIEPOINT][:6])
            self.left, self.top, self._xres, self._yres = x - i * sx, y + j * sy, sx, sy
        elif _TRANSFORM in t:
            m = [float(v) for v in t[_TRANSFORM]]
            if m[1] != 0.0 or m[4] != 0.0 or m[5] >= 0.0:
                raise GeoTiffError(f"{path}: rotated / south-up rasters are not supported")
            self.left, self.top, self._xres, self._yres = m[3], m[7], m[0], -m[5]
        else:
            self.left, self.top, self._xres, self._yres = 0.0, float(self.height), 1.0, 1.0  # rasterio's identity
        self.geokeys = tuple(int(v) for v in t.get(_GEOKEYS, ()))
        self.geoascii = t.get(_GEOASCII, ("",))[0]
        self.geodoubles = tuple(float(v) for v in t.get(_GEODOUBLES, ()))
        epsg = _epsg_from_geokeys(self.geokeys)
        self.crs = f"EPSG:{epsg}" if epsg else None
        nd = t.get(_NODATA, (None,))[0]
        try:
            self.nodata = float(nd) if nd not in (None, "") else None
        except ValueError:
            self.nodata = None
        self.closed = False

    # ---- rasterio-like attributes -------------------------------------------------------------------------
    @property
    def res(self) -> Tuple[float, float]:
        return (self._xres, self._yres)

    @property
    def transform(self) -> Affine:
        return Affine(self._xres, 0.0, self.left, 0.0, -self._yres, self.top)

    @property
    def dtypes(self) -> Tuple[str, ...]:
        return (str(self._native),) * self.count

    @property
    def profile(self) -> dict:
        return {"driver": "GTiff", "height": self.height, "width": self.width, "count": self.count,
                "dtype": str(self._native), "crs": self.crs, "transform": self.transform, "nodata": self.nodata,
                "tiled": _TW in self.tags, "blockxsize": self._bw, "blockysize": self._bh,
                "compress": {1: None, 5: "lzw"}.get(self._comp, "deflate"),
                "interleave": "band" if self._planar == 2 else "pixel"}

    # ---- block access -------------------------------------------------------------------------------------
    def _load_block(self, idx: int) -> np.ndarray:
        """Block ``idx`` as a native-endian array [bh, bw, samples] (samples = count, or 1 for band-separate);
        touches no shared state, so several blocks decode in parallel (the codecs release the GIL)"""
        spp = 1 if self._planar == 2 else self.count
        shape = (self._bh, self._bw, spp)
        nbytes = self._bh * self._bw * spp * self.dtype.itemsize
        off, cnt = int(self._offs[idx]), int(self._cnts[idx])
        if _TW not in self.tags:  # the last strip holds only the remaining rows
            by = (idx % (self._nbx * self._nby)) // self._nbx
            rows = min(self._bh, self.height - by * self._bh)
            shape = (rows, self._bw, spp)
            nbytes = rows * self._bw * spp * self.dtype.itemsize
        if cnt == 0:  # sparse file (GDAL SPARSE_OK): a block that was never written reads as zeros
            arr = np.zeros(shape, self._native)
        else:
            if off + cnt > len(self._mm):
                raise GeoTiffError(f"{self.path}: block {idx} lies outside the file")
            raw = self._mm[off:off + cnt]
            if self._comp == 1:
                if cnt < nbytes:
                    raise GeoTiffError(f"{self.path}: block {idx} is truncated")
                data = np.frombuffer(raw, np.uint8, nbytes)
            elif self._comp == 5:
                data = np.empty(nbytes, np.uint8)
                src = np.frombuffer(raw, np.uint8)
                n = _codec().ffa_tiff_lzw_decode(src.ctypes.data, src.size, data.ctypes.data, nbytes)
                if n != nbytes:
                    from flairhip import lib
                    raise GeoTiffError(f"{self.path}: LZW block {idx} decoded to {n} of {nbytes} bytes "
                                       f"({lib.load().ffa_last_error().decode('utf-8', 'replace') if n < 0 else 'short'})")
            else:
                dec = zlib.decompress(raw)
                if len(dec) < nbytes:
                    raise GeoTiffError(f"{self.path}: Deflate block {idx} is truncated")
                data = np.frombuffer(dec, np.uint8, nbytes)
            arr = data.view(self.dtype).reshape(shape)
            if arr.dtype != self._native or not arr.flags.writeable:
                arr = arr.astype(self._native)  # byte swap and / or private copy
            if self._pred == 2:
                rc = _codec().ffa_tiff_hpredict(arr.ctypes.data, shape[0], shape[1] * spp, self.dtype.itemsize, spp, 1)
                if rc != 0:
                    raise GeoTiffError(f"{self.path}: predictor failed on block {idx}")
        return arr

    def _blocks(self, ids: Sequence[int]) -> Dict[int, np.ndarray]:
        """decoded blocks by index: cache hits, the rest decoded (in parallel when several are missing) and cached"""
        got: Dict[int, np.ndarray] = {}
        missing = []
        for i in ids:
            a = self._cache.get(i)
            if a is not None:
                self._cache.move_to_end(i)
                got[i] = a
            elif i not in got:
                got[i] = None
                missing.append(i)
        if missing:
            compressed = self._comp != 1
            if compressed and len(missing) > 3:
                # a handful of blocks per task: a pool task costs ~30 us of Python, a 64 KiB block ~300 us to decode
                nw = min(_POOL_WORKERS, len(missing) // 2)
                parts = [missing[i::nw] for i in range(nw)]
                done = _pool().map(lambda ids_: [self._load_block(i) for i in ids_], parts)
                by_id = {i: a for ids_, arrs in zip(parts, done) for i, a in zip(ids_, arrs)}
                loaded = [by_id[i] for i in missing]
            else:
                loaded = [self._load_block(i) for i in missing]
            for i, a in zip(missing, loaded):
                got[i] = a
                self._cache[i] = a
                self._cache_bytes += a.nbytes
            while self._cache_bytes > self._cache_cap and len(self._cache) > 1:
                _, old = self._cache.popitem(last=False)
                self._cache_bytes -= old.nbytes
        return got

    def _block(self, bands: Sequence[int], ys: int, ye: int, xs: int, xe: int, out: np.ndarray) -> None:
        """out[k] = band bands[k] (0-based) over rows ys..ye, columns xs..xe (all inside the raster)"""
        per_plane = self._nbx * self._nby
        planes = list(bands) if self._planar == 2 else [0]
        cells = [(by, bx) for by in range(ys // self._bh, (ye - 1) // self._bh + 1)
                 for bx in range(xs // self._bw, (xe - 1) // self._bw + 1)]
        blocks = self._blocks([pl * per_plane + by * self._nbx + bx for pl in planes for by, bx in cells])
        for by, bx in cells:
            y0, x0 = by * self._bh, bx * self._bw
            a, b = max(ys, y0), min(ye, y0 + self._bh)
            c, d = max(xs, x0), min(xe, x0 + self._bw)
            for k, band in enumerate(bands):
                if self._planar == 2:
                    blk, ch = blocks[band * per_plane + by * self._nbx + bx], 0
                else:
                    blk, ch = blocks[by * self._nbx + bx], band
                out[k, a - ys:b - ys, c - xs:d - xs] = blk[a - y0:b - y0, c - x0:d - x0, ch]

    def close(self) -> None:
        if getattr(self, "closed", False):
            return
        self.closed = True
        self._cache = OrderedDict()
        try:
            if getattr(self, "_mm", None) is not None:
                self._mm.close()
        finally:
            self._f.close()


WRITTEN_SUFFIX = ".written.tif"  # sidecar mask of a sharded run's part file


class GeoTiffWriter(ArrayRaster):
    """uint8 (or any integer / float dtype) GeoTIFF written on close(): tiled 256 x 256, LZW, band-separate planes for
    multi-band rasters, BigTIFF when the file would pass 4 GB.  Behaves as an ArrayRaster until then."""

    BLOCK = 256

    def __init__(self, path: str, width: int, height: int, count: int, left: float, top: float, res,
                 crs: Optional[str] = None, dtype=np.uint8, compress: Optional[str] = "lzw", geokeys: Sequence[int] = (),
                 geoascii: str = "", geodoubles: Sequence[float] = (), nodata=None, scratch_above: int = 2 << 30):
        nbytes = int(count) * int(height) * int(width) * np.dtype(dtype).itemsize
        self._scratch = None
        if nbytes > scratch_above:  # keep very large class-probability rasters out of RAM
            fd, self._scratch = tempfile.mkstemp(suffix=".raw", dir=os.path.dirname(os.path.abspath(path)) or ".")
            os.close(fd)
            data = np.memmap(self._scratch, dtype=dtype, mode="w+", shape=(count, height, width))
        else:
            data = np.zeros((count, height, width), dtype=dtype)
        xres, yres = (res if isinstance(res, (tuple, list)) else (res, res))
        super().__init__(data, left, top, float(xres), crs)
        self._yres = float(yres)
        self.path = path
        if compress not in (None, "none", "lzw", "deflate", "LZW", "DEFLATE", "NONE"):
            raise GeoTiffError(f"compress={compress!r} is not supported (none / lzw / deflate are)")
        self.compress = (compress or "none").lower()
        self.geokeys = tuple(int(v) for v in geokeys)
        self.geoascii, self.geodoubles, self.nodata = geoascii, tuple(geodoubles), nodata
        if not self.geokeys and crs and str(crs).upper().startswith("EPSG:"):
            self.geokeys = _geokeys_for_epsg(int(str(crs).split(":")[1]))

    @classmethod
    def like(cls, path: str, ref, count: int, dtype=np.uint8, width: Optional[int] = None, height: Optional[int] = None,
             left: Optional[float] = None, top: Optional[float] = None, res=None, compress="lzw") -> "GeoTiffWriter":
        """Output raster on the grid of ``ref`` (a GeoTiffRaster / ArrayRaster), optionally re-gridded"""
        b = ref.bounds
        return cls(path, width if width is not None else ref.width, height if height is not None else ref.height,
                   count, b.left if left is None else left, b.top if top is None else top,
                   ref.res if res is None else res, crs=getattr(ref, "crs", None), dtype=dtype, compress=compress,
                   geokeys=getattr(ref, "geokeys", ()), geoascii=getattr(ref, "geoascii", ""),
                   geodoubles=getattr(ref, "geodoubles", ()))

    @property
    def res(self) -> Tuple[float, float]:
        return (self._res, self._yres)

    @property
    def bounds(self) -> BoundingBox:
        return BoundingBox(self.left, self.top - self.height * self._yres, self.left + self.width * self._res, self.top)

    @property
    def profile(self) -> dict:
        return {"driver": "GTiff", "height": self.height, "width": self.width, "count": self.count,
                "dtype": str(self.data.dtype), "crs": self.crs, "compress": self.compress, "tiled": True,
                "blockxsize": self.BLOCK, "blockysize": self.BLOCK,
                "transform": Affine(self._res, 0.0, self.left, 0.0, -self._yres, self.top)}

    def _encode(self, job) -> bytes:
        band, by, bx = job
        B = self.BLOCK
        blk = np.zeros((B, B), self.data.dtype)
        src = self.data[band, by * B:(by + 1) * B, bx * B:(bx + 1) * B]
        blk[:src.shape[0], :src.shape[1]] = src
        if self.compress == "none":
            return blk.tobytes()
        if self.compress == "deflate":
            return zlib.compress(blk.tobytes(), 6)
        L = _codec()
        flat = blk.reshape(-1).view(np.uint8)
        cap = L.ffa_tiff_lzw_bound(flat.size)
        out = np.empty(cap, np.uint8)
        n = L.ffa_tiff_lzw_encode(flat.ctypes.data, flat.size, out.ctypes.data, cap)
        if n <= 0:
            raise GeoTiffError(f"LZW encode failed ({n})")
        return out[:n].tobytes()

    def close(self) -> None:
        if self.closed:
            return
        self.closed = True
        B = self.BLOCK
        nbx, nby = -(-self.width // B), -(-self.height // B)
        jobs = [(band, by, bx) for band in range(self.count) for by in range(nby) for bx in range(nbx)]
        item = self.data.dtype.itemsize
        big = self.data.nbytes > (2 << 30)  # LZW can expand noise by 1.5x: stay clear of the 4 GB offsets
        kind = {"u": 1, "i": 2, "f": 3}[self.data.dtype.kind]
        offs: List[int] = []
        cnts: List[int] = []
        tmp = self.path + ".part"
        with open(tmp, "wb") as f:
            f.write(b"II" + (struct.pack("<HHHQ", 43, 8, 0, 0) if big else struct.pack("<HI", 42, 0)))
            pos = f.tell()
            for lo in range(0, len(jobs), 1024):  # bounded number of encoded blocks in flight
                for enc in _pool().map(self._encode, jobs[lo:lo + 1024]):
                    f.write(enc)
                    offs.append(pos)
                    cnts.append(len(enc))
                    pos += len(enc)
                    if pos & 1:
                        f.write(b"\0")
                        pos += 1
            if not big and pos > (1 << 32) - (1 << 20):
                raise GeoTiffError("classic TIFF overflow: the compressed raster passed 4 GB")
            # ---- the IFD, after the pixel data ----
            LONGT = 16 if big else 4
            entries = [(_W, 4, (self.width,)), (_H, 4, (self.height,)), (_BITS, 3, (item * 8,) * self.count),
                       (_COMP, 3, ({"none": 1, "lzw": 5, "deflate": 8}[self.compress],)), (_PHOTO, 3, (1,)),
                       (_SPP, 3, (self.count,)), (_PLANAR, 3, (2 if self.count > 1 else 1,)),
                       (_TW, 3, (B,)), (_TL, 3, (B,)), (_TILE_OFF, LONGT, tuple(offs)), (_TILE_CNT, LONGT, tuple(cnts)),
                       (_FORMAT, 3, (kind,) * self.count),
                       (_PIXSCALE, 12, (self._res, self._yres, 0.0)),
                       (_TIEPOINT, 12, (0.0, 0.0, 0.0, self.left, self.top, 0.0))]
            if self.count > 1:
                entries.append((_EXTRA, 3, (0,) * (self.count - 1)))
            if self.geokeys:
                entries.append((_GEOKEYS, 3, self.geokeys))
            if self.geodoubles:
                entries.append((_GEODOUBLES, 12, self.geodoubles))
            if self.geoascii:
                entries.append((_GEOASCII, 2, self.geoascii))
            if self.nodata is not None:
                entries.append((_NODATA, 2, repr(self.nodata) if isinstance(self.nodata, float) else str(self.nodata)))
            entries.sort(key=lambda e: e[0])
            inl, ofmt, cfmt, esz = (8, "Q", "Q", 20) if big else (4, "I", "I", 12)
            ifd_off = pos
            table = 8 + len(entries) * esz + 8 if big else 2 + len(entries) * esz + 4
            extra_pos = ifd_off + table
            body, extra = b"", b""
            for tag, typ, vals in entries:
                if typ == 2:
                    payload = vals.encode("latin-1") + b"\0"
                    n = len(payload)
                else:
                    payload = struct.pack("<" + _TYPE[typ][0] * len(vals), *vals)
                    n = len(vals)
                if len(payload) <= inl:
                    field = payload.ljust(inl, b"\0")
                else:
                    field = struct.pack("<" + ofmt, extra_pos + len(extra))
                    extra += payload + (b"\0" if len(payload) & 1 else b"")
                body += struct.pack("<HH" + cfmt, tag, typ, n) + field
            f.write((struct.pack("<Q", len(entries)) if big else struct.pack("<H", len(entries))) + body +
                    struct.pack("<" + ofmt, 0) + extra)
            f.seek(8 if big else 4)
            f.write(struct.pack("<" + ofmt, ifd_off))
        os.replace(tmp, self.path)
        if self.written is not None:  # sharded run: the pixels this rank wrote, for merge_shard_files
            m = GeoTiffWriter.like(self.path + WRITTEN_SUFFIX, self, 1)
            m.data[0] = self.written
            m.close()
        if self._scratch is not None:
            del self.data
            os.unlink(self._scratch)
            self._scratch = None
            self.data = np.zeros((0, 0, 0), np.uint8)


def merge_shard_files(part_paths: Sequence[str], out_path: str, rows_per_pass: int = 2048) -> str:
    """One prediction raster from the part files of a sharded zonal run (one process per GPU, each wrote
    ``<part>`` + ``<part>.written.tif``).  Parts are applied in the order given (= rank order, which is tile order:
    where the clamped last row / column re-covers pixels the later tile wins, as in the reference's single loop,
    inference.py:349).  Works through the rasters ``rows_per_pass`` rows at a time."""
    parts = [GeoTiffRaster(p) for p in part_paths]
    masks = [GeoTiffRaster(p + WRITTEN_SUFFIX) for p in part_paths]
    try:
        first = parts[0]
        for r in parts[1:] + masks:
            if r.shape != first.shape:
                raise GeoTiffError(f"{r.path}: shape {r.shape} differs from {first.path}: {first.shape}")
        from flair_zonal_detection.raster import make_window
        out = GeoTiffWriter.like(out_path, first, first.count, dtype=first._native)
        for y in range(0, first.height, rows_per_pass):
            h = min(rows_per_pass, first.height - y)
            win = make_window(0, y, first.width, h)
            dst = out.data[:, y:y + h]
            for r, m in zip(parts, masks):
                sel = m.read(1, window=win).astype(bool)
                if sel.any():
                    dst[:, sel] = r.read(window=win)[:, sel]
        out.close()
    finally:
        for r in parts + masks:
            r.close()
    return out_path
