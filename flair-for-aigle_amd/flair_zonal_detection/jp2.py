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
  plugin, ONCE per raster on the first window read, into host memory ([bands, H, W]; a 25 000 x 25 000 RGB mosaic is
  1.9 GB, which the 270 GB host budget of a GPU box holds many times over).  Pillow exposes no region decode, so the
  per-tile reads of the loop are slices of the resident array; the duck-typed surface is ``RasterBase``'s
  (read / read_bounds / bounds / res / count / dtypes / profile / crs), the same as ``GeoTiffRaster``'s.

Raises ``Jp2Error`` with the reason when Pillow was built without OpenJPEG or the file is not JPEG-2000.
"""
from __future__ import annotations

import os
import struct
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
        with open(path, "rb") as f:
            if f.read(12) == JP2_SIGNATURE:
                for tbox, off, n in iter_boxes(f):
                    if tbox == b"uuid" and n >= 16:
                        f.seek(off)
                        if f.read(16) == GEOJP2_UUID:
                            geo = _georef_from_tiff(f.read(n - 16))
                            if geo is not None:
                                break
                    elif tbox == b"jp2c":
                        break  # georeferencing boxes precede the code-stream in every writer's layout; stop scanning
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

    def _block(self, bands, ys: int, ye: int, xs: int, xe: int, out: np.ndarray) -> None:
        data = self._decoded()
        for k, bnd in enumerate(bands):
            out[k] = data[bnd, ys:ye, xs:xe]

    def close(self) -> None:
        self._data = None
        self.closed = True
