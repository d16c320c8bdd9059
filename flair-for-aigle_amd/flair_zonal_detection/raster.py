"""Raster access for the zonal loop.

The reference talks to rasterio (GDAL) directly (flair_zonal_detection/dataset.py:89-117 windowed reads,
inference.py:157-208 / :342-352 GeoTIFF window writes).  rasterio is not installed in the build image, so the
loop is written against the small duck-typed surface below: a real ``rasterio`` dataset satisfies it,
``ArrayRaster`` is the in-memory raster (north-up, square pixels), and ``geotiff.GeoTiffRaster`` /
``geotiff.GeoTiffWriter`` / ``jp2.Jp2Raster`` are the file-backed ones.
"""
from __future__ import annotations

import os
from collections import namedtuple
from typing import Optional, Tuple

import numpy as np

BoundingBox = namedtuple("BoundingBox", ["left", "bottom", "right", "top"])
Window = namedtuple("Window", ["col_off", "row_off", "width", "height"])


_RIO_WINDOW = None  # rasterio.windows.Window, or False once the import has failed (a failed import costs ~70 us)


def make_window(col_off: int, row_off: int, width: int, height: int):
    """rasterio.windows.Window when rasterio is importable (so real datasets accept it), else the tuple above."""
    global _RIO_WINDOW
    if _RIO_WINDOW is None:
        try:
            from rasterio.windows import Window as RioWindow  # type: ignore
            _RIO_WINDOW = RioWindow
        except ImportError:
            _RIO_WINDOW = False
    if _RIO_WINDOW:
        return _RIO_WINDOW(col_off=col_off, row_off=row_off, width=width, height=height)
    return Window(col_off, row_off, width, height)


class RasterBase:
    """Window / bounds reads shared by the in-memory and the GeoTIFF rasters.  A subclass provides ``left``, ``top``,
    ``width``, ``height``, ``count``, ``res`` = (xres, yres), ``dtype`` and ``_block(bands, ys, ye, xs, xe, out)``
    (copy 0-based ``bands`` over rows ys..ye / columns xs..xe, all inside the raster, into out[k])."""

    @property
    def shape(self) -> Tuple[int, int]:
        return (self.height, self.width)

    @property
    def bounds(self) -> BoundingBox:
        xres, yres = self.res
        return BoundingBox(self.left, self.top - self.height * yres, self.left + self.width * xres, self.top)

    def _native_dtype(self) -> np.dtype:
        return np.dtype(self.dtype).newbyteorder("=")

    def prefetch_bounds(self, bounds) -> None:
        """Hint: the geographic box (left, bottom, right, top) is about to be read tile by tile.  Rasters whose storage
        blocks are expensive to decode (JPEG-2000 code-stream tiles) decode the blocks under it in parallel; the default
        does nothing."""

    def read_bounds(self, indexes, bounds, out_size: int, out: np.ndarray = None, nearest: bool = False) -> np.ndarray:
        """Boundless read of the geographic box `bounds` = (left, bottom, right, top), zero fill outside the
        raster, resampled to out_size x out_size when the box is not already that many pixels (bilinear; ``nearest``:
        the source pixel under each output pixel centre, as Resampling.nearest does for the cloud masks).
        ``out`` ([len(indexes), out_size, out_size], the raster's dtype) receives the tile in place when given
        (the zonal loop passes a slot of its pinned batch buffer)."""
        l, b, r, t = bounds
        xres, yres = self.res
        c0 = (l - self.left) / xres
        r0 = (self.top - t) / yres
        w = (r - l) / xres
        h = (t - b) / yres
        bands = [i - 1 for i in indexes]
        dtype = self._native_dtype()
        ci, ri, wi, hi = int(round(c0)), int(round(r0)), int(round(w)), int(round(h))
        if abs(c0 - ci) < 1e-6 and abs(r0 - ri) < 1e-6 and wi == out_size and hi == out_size:
            ys, ye = max(ri, 0), min(ri + hi, self.height)
            xs, xe = max(ci, 0), min(ci + wi, self.width)
            inside = ys == ri and xs == ci and ye == ri + hi and xe == ci + wi
            if out is None:
                out = np.empty((len(bands), out_size, out_size), dtype=dtype)
            if not inside:
                out[...] = 0
            if ye > ys and xe > xs:
                self._block(bands, ys, ye, xs, xe, out[:, ys - ri:ye - ri, xs - ci:xe - ci])
            return out
        if nearest:
            yy = np.floor(r0 + (np.arange(out_size) + 0.5) * (h / out_size)).astype(int)
            xx = np.floor(c0 + (np.arange(out_size) + 0.5) * (w / out_size)).astype(int)
            res_ = np.zeros((len(bands), out_size, out_size), dtype=dtype)
            oky, okx = (yy >= 0) & (yy < self.height), (xx >= 0) & (xx < self.width)
            if oky.any() and okx.any():
                ya, yb, xa, xb = int(yy[oky].min()), int(yy[oky].max()) + 1, int(xx[okx].min()), int(xx[okx].max()) + 1
                blk = np.empty((len(bands), yb - ya, xb - xa), dtype=dtype)
                self._block(bands, ya, yb, xa, xb, blk)
                res_[:, np.where(oky)[0][:, None], np.where(okx)[0][None, :]] = blk[:, (yy[oky] - ya)[:, None], (xx[okx] - xa)[None, :]]
            if out is not None:
                out[...] = res_
                return out
            return res_
        # generic path: bilinear sampling at output pixel centres, from the sub-block the samples touch
        ys = r0 + (np.arange(out_size) + 0.5) * (h / out_size) - 0.5
        xs = c0 + (np.arange(out_size) + 0.5) * (w / out_size) - 0.5
        y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
        fy, fx = (ys - y0)[None, :, None], (xs - x0)[None, None, :]
        ya, yb = max(int(y0.min()), 0), min(int(y0.max()) + 2, self.height)
        xa, xb = max(int(x0.min()), 0), min(int(x0.max()) + 2, self.width)
        if yb > ya and xb > xa:
            blk = np.empty((len(bands), yb - ya, xb - xa), dtype=dtype)
            self._block(bands, ya, yb, xa, xb, blk)
            src = blk.astype(np.float32)
        else:  # the box lies outside the raster
            ya, yb, xa, xb = 0, 1, 0, 1
            src = np.zeros((len(bands), 1, 1), np.float32)

        def at(yy, xx):
            ok = ((yy >= 0) & (yy < self.height))[:, None] & ((xx >= 0) & (xx < self.width))[None, :]
            v = src[:, np.clip(yy, ya, yb - 1) - ya][:, :, np.clip(xx, xa, xb - 1) - xa]
            return v * ok[None]

        out_ = (at(y0, x0) * (1 - fy) * (1 - fx) + at(y0, x0 + 1) * (1 - fy) * fx +
                at(y0 + 1, x0) * fy * (1 - fx) + at(y0 + 1, x0 + 1) * fy * fx)
        res_ = out_.astype(dtype) if np.issubdtype(dtype, np.floating) else out_
        if out is not None:
            # integer destination: round to nearest like GDAL's resampler (a plain cast would truncate)
            out[...] = res_ if np.issubdtype(out.dtype, np.floating) else np.rint(res_)
            return out
        return res_

    def read(self, indexes=None, window=None, boundless: bool = False, fill_value=0, out_shape=None, **_ignored):
        """rasterio's ``read`` for pixel windows: [n, h, w] (or [h, w] for a single int index); ``boundless`` fills
        the part of the window outside the raster with ``fill_value``.  Resampling reads go through read_bounds."""
        single = isinstance(indexes, int)
        idx = [indexes] if single else (list(indexes) if indexes is not None else list(range(1, self.count + 1)))
        if any(i < 1 or i > self.count for i in idx):
            raise IndexError(f"band index out of range 1..{self.count}: {idx}")
        if window is None:
            c, r, w, h = 0, 0, self.width, self.height
        else:
            c, r, w, h = int(window.col_off), int(window.row_off), int(window.width), int(window.height)
        if out_shape is not None and tuple(out_shape)[-2:] != (h, w):
            raise NotImplementedError("resampling reads: use read_bounds(indexes, bounds, out_size)")
        ys, ye, xs, xe = max(r, 0), min(r + h, self.height), max(c, 0), min(c + w, self.width)
        if not boundless:
            r, c, h, w = ys, xs, max(ye - ys, 0), max(xe - xs, 0)
        out = np.full((len(idx), h, w), fill_value, dtype=self._native_dtype())
        if ye > ys and xe > xs:
            self._block([i - 1 for i in idx], ys, ye, xs, xe, out[:, ys - r:ye - r, xs - c:xe - c])
        return out[0] if single else out

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class ArrayRaster(RasterBase):
    """[count, H, W] array with a north-up geotransform; mimics the rasterio dataset attributes the loop reads
    (.bounds, .res, .shape, .height, .width, .count, .profile, .crs) plus read / write by window."""

    def __init__(self, data: np.ndarray, left: float, top: float, res: float, crs: Optional[str] = "EPSG:2154"):
        if data.ndim == 2:
            data = data[None]
        self.data = data
        self.left, self.top, self._res, self.crs = float(left), float(top), float(res), crs
        self.closed = False
        self.written = None  # optional [H, W] bool mask of the pixels write() touched (sharded runs merge by it)

    def track_writes(self) -> "ArrayRaster":
        self.written = np.zeros((self.height, self.width), dtype=bool)
        return self

    @classmethod
    def empty_like(cls, ref: "ArrayRaster", count: int, dtype=np.uint8) -> "ArrayRaster":
        return cls(np.zeros((count, ref.height, ref.width), dtype=dtype), ref.left, ref.top, ref._res, ref.crs)

    @property
    def count(self) -> int:
        return self.data.shape[0]

    @property
    def height(self) -> int:
        return self.data.shape[1]

    @property
    def width(self) -> int:
        return self.data.shape[2]

    @property
    def dtype(self) -> np.dtype:
        return self.data.dtype

    @property
    def res(self) -> Tuple[float, float]:
        return (self._res, self._res)

    @property
    def profile(self) -> dict:
        return {"driver": "MEM", "height": self.height, "width": self.width, "count": self.count,
                "dtype": str(self.data.dtype), "crs": self.crs}

    def _block(self, bands, ys: int, ye: int, xs: int, xe: int, out: np.ndarray) -> None:
        for k, bnd in enumerate(bands):  # plain slices: no fancy-index temporary
            out[k] = self.data[bnd, ys:ye, xs:xe]

    def write(self, arr: np.ndarray, band: int, window=None) -> None:
        if window is None:
            self.data[band - 1] = arr
            return
        c, r, w, h = int(window.col_off), int(window.row_off), int(window.width), int(window.height)
        arr = np.asarray(arr).reshape(h, w)
        # The reference clips windows on the right / bottom only (inference.py:328-335); a raster smaller than one
        # tile puts the clamped tile's kept area at a negative offset, which GDAL refuses.  Clip on every side here.
        y0, x0 = max(-r, 0), max(-c, 0)
        y1, x1 = min(h, self.height - r), min(w, self.width - c)
        if y1 <= y0 or x1 <= x0:
            return
        self.data[band - 1, r + y0:r + y1, c + x0:c + x1] = arr[y0:y1, x0:x1]
        if self.written is not None:
            self.written[r + y0:r + y1, c + x0:c + x1] = True

    def close(self) -> None:
        self.closed = True


def open_raster(path_or_raster):
    """A raster object for a path, or the object itself when it already is one.  rasterio (GDAL) opens the path when
    it is installed; without it GeoTIFF files are read by flair_zonal_detection.geotiff and JPEG-2000 mosaics by
    flair_zonal_detection.jp2 (other GDAL-only formats raise)."""
    if not isinstance(path_or_raster, (str, bytes, os.PathLike)):
        return path_or_raster
    try:
        import rasterio  # type: ignore
    except ImportError:
        from flair_zonal_detection import jp2
        path = os.fspath(path_or_raster)
        if jp2.is_jpeg2000(path):
            return jp2.Jp2Raster(path)
        from flair_zonal_detection.geotiff import GeoTiffRaster
        return GeoTiffRaster(path)
    return rasterio.open(path_or_raster)
