"""Raster access for the zonal loop.

The reference talks to rasterio (GDAL) directly (flair_zonal_detection/dataset.py:89-117 windowed reads,
inference.py:157-208 / :342-352 GeoTIFF window writes).  GDAL-backed I/O is outside this round's scope
 and rasterio is not installed in the build image, so the loop is written
against the small duck-typed surface below: a real ``rasterio`` dataset satisfies it, and ``ArrayRaster``
is an in-memory stand-in (north-up, square pixels) that lets the tile loop run end to end.
"""
from __future__ import annotations

from collections import namedtuple
from typing import Optional, Tuple

import numpy as np

BoundingBox = namedtuple("BoundingBox", ["left", "bottom", "right", "top"])
Window = namedtuple("Window", ["col_off", "row_off", "width", "height"])


def make_window(col_off: int, row_off: int, width: int, height: int):
    """rasterio.windows.Window when rasterio is importable (so real datasets accept it), else the tuple above."""
    try:
        from rasterio.windows import Window as RioWindow  # type: ignore
        return RioWindow(col_off=col_off, row_off=row_off, width=width, height=height)
    except ImportError:
        return Window(col_off, row_off, width, height)


class ArrayRaster:
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
    def shape(self) -> Tuple[int, int]:
        return (self.height, self.width)

    @property
    def res(self) -> Tuple[float, float]:
        return (self._res, self._res)

    @property
    def bounds(self) -> BoundingBox:
        return BoundingBox(self.left, self.top - self.height * self._res, self.left + self.width * self._res, self.top)

    @property
    def profile(self) -> dict:
        return {"driver": "MEM", "height": self.height, "width": self.width, "count": self.count,
                "dtype": str(self.data.dtype), "crs": self.crs}

    def read_bounds(self, indexes, bounds, out_size: int, out: np.ndarray = None) -> np.ndarray:
        """Boundless read of the geographic box `bounds` = (left, bottom, right, top), zero fill outside the
        raster, resampled to out_size x out_size when the box is not already that many pixels (bilinear).
        ``out`` ([len(indexes), out_size, out_size], the raster's dtype) receives the tile in place when given
        (the zonal loop passes a slot of its pinned batch buffer)."""
        l, b, r, t = bounds
        c0 = (l - self.left) / self._res
        r0 = (self.top - t) / self._res
        w = (r - l) / self._res
        h = (t - b) / self._res
        bands = [i - 1 for i in indexes]
        ci, ri, wi, hi = int(round(c0)), int(round(r0)), int(round(w)), int(round(h))
        if abs(c0 - ci) < 1e-6 and abs(r0 - ri) < 1e-6 and wi == out_size and hi == out_size:
            ys, ye = max(ri, 0), min(ri + hi, self.height)
            xs, xe = max(ci, 0), min(ci + wi, self.width)
            inside = ys == ri and xs == ci and ye == ri + hi and xe == ci + wi
            if out is None:
                out = np.empty((len(bands), out_size, out_size), dtype=self.data.dtype)
            if not inside:
                out[...] = 0
            if ye > ys and xe > xs:
                for k, bnd in enumerate(bands):  # plain slices: no fancy-index temporary
                    out[k, ys - ri:ye - ri, xs - ci:xe - ci] = self.data[bnd, ys:ye, xs:xe]
            return out
        # generic path: bilinear sampling at output pixel centres
        ys = r0 + (np.arange(out_size) + 0.5) * (h / out_size) - 0.5
        xs = c0 + (np.arange(out_size) + 0.5) * (w / out_size) - 0.5
        y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
        fy, fx = (ys - y0)[None, :, None], (xs - x0)[None, None, :]
        src = self.data[bands].astype(np.float32)

        def at(yy, xx):
            ok = ((yy >= 0) & (yy < self.height))[:, None] & ((xx >= 0) & (xx < self.width))[None, :]
            v = src[:, np.clip(yy, 0, self.height - 1)][:, :, np.clip(xx, 0, self.width - 1)]
            return v * ok[None]

        out_ = (at(y0, x0) * (1 - fy) * (1 - fx) + at(y0, x0 + 1) * (1 - fy) * fx +
                at(y0 + 1, x0) * fy * (1 - fx) + at(y0 + 1, x0 + 1) * fy * fx)
        res_ = out_.astype(self.data.dtype) if np.issubdtype(self.data.dtype, np.floating) else out_
        if out is not None:
            out[...] = res_
            return out
        return res_

    def write(self, arr: np.ndarray, band: int, window=None) -> None:
        if window is None:
            self.data[band - 1] = arr
            return
        c, r, w, h = int(window.col_off), int(window.row_off), int(window.width), int(window.height)
        self.data[band - 1, r:r + h, c:c + w] = arr
        if self.written is not None:
            self.written[r:r + h, c:c + w] = True

    def close(self) -> None:
        self.closed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def open_raster(path_or_raster):
    """A raster object for a path (needs rasterio) or the object itself when it already is one."""
    if not isinstance(path_or_raster, (str, bytes)):
        return path_or_raster
    try:
        import rasterio  # type: ignore
    except ImportError as e:
        raise ImportError("opening raster files needs rasterio (GDAL), which is outside this build's scope; pass an "
                          "ArrayRaster or any rasterio-like object instead") from e
    return rasterio.open(path_or_raster)
