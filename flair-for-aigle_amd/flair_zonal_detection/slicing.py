"""Sliding-window tile grid -- counterpart of the reference's flair_zonal_detection/slicing.py
(generate_patches_from_reference :20-121).

The float64 bookkeeping (np.arange semantics, clamping of the last row / column, kept-area bounds, the
round(.., 6) duplicate filter and the "1-{row}-{col}" ids) is done by ``ffa_slice_grid`` in libflairhip
(csrc/tile_grid.cpp), bit-exact with the reference; tests pin it against grids produced by the reference
itself (tests/golden/slicing_*.json).  This module supplies the reference's function signature and
return type around it: a (Geo)DataFrame with the same columns, in the same order (x outer, y inner).
"""
from __future__ import annotations

import logging
import math
import os
from typing import Dict, Optional, Sequence, Tuple

import pandas as pd

from flairhip import ops
from flair_zonal_detection.raster import open_raster

logger = logging.getLogger(__name__)

COLUMNS = ["id", "input_id", "output_id", "job_done", "left", "bottom", "right", "top", "left_o", "bottom_o",
           "right_o", "top_o", "geometry"]


def create_box_from_bounds(x_min: float, x_max: float, y_min: float, y_max: float):
    """shapely box when shapely is importable, else the (minx, miny, maxx, maxy) tuple."""
    try:
        from shapely.geometry import box  # type: ignore
        return box(x_min, y_max, x_max, y_min)
    except ImportError:
        return (x_min, y_min, x_max, y_max)


def _zone_bounds(src, geozone) -> Optional[Tuple[float, float, float, float]]:
    """Bounds of (raster INTERSECT geozone), snapped outward to the raster's pixel grid like
    rasterio.mask.mask(src, geozone, crop=True) does (slicing.py:42-48); None when they do not intersect."""
    rb = src.bounds
    res = src.res[0]
    if geozone is None:
        return (rb.left, rb.bottom, rb.right, rb.top)
    if hasattr(geozone, "bounds"):
        gb = geozone.bounds
    elif isinstance(geozone, Sequence) and len(geozone) == 4 and all(isinstance(v, (int, float)) for v in geozone):
        gb = tuple(geozone)
    else:  # iterable of geometries
        bs = [g.bounds for g in geozone]
        gb = (min(b[0] for b in bs), min(b[1] for b in bs), max(b[2] for b in bs), max(b[3] for b in bs))
    h, w = src.shape[0], src.shape[1]
    c0 = max(0, math.floor((gb[0] - rb.left) / res))
    c1 = min(w, math.ceil((gb[2] - rb.left) / res))
    r0 = max(0, math.floor((rb.top - gb[3]) / res))
    r1 = min(h, math.ceil((rb.top - gb[1]) / res))
    if c1 <= c0 or r1 <= r0:
        return None
    # rasterio.transform.array_bounds(height, width, out_transform) of the cropped array
    left = rb.left + c0 * res
    top = rb.top - r0 * res
    return (left, top - (r1 - r0) * res, left + (c1 - c0) * res, top)


def slice_bounds(zone_bounds, ref_bounds, patch_size: int, margin: int, resolution: float):
    """The reference's core loop (slicing.py:51-112) on plain bounds -> list of libflairhip Tile records."""
    min_x, min_y, max_x, max_y = zone_bounds
    return ops.slice_grid(min_x, min_y, max_x, max_y, ref_bounds[0], ref_bounds[1], patch_size, margin, resolution)


def generate_patches_from_reference(config: Dict, img_path, geozone_contour_geometries=None):
    """Slice the reference raster (intersected with the geozone) into overlapping tiles.

    ``img_path`` is a raster path (rasterio required) or a raster-like object (flair_zonal_detection.raster).
    Returns a GeoDataFrame when geopandas is importable, else a pandas DataFrame with the same columns.
    """
    patch_size = config["img_pixels_detection"]
    margin = config["margin"]
    output_name = config["output_name"]
    resolution = config["reference_resolution"]

    src = open_raster(img_path)
    try:
        zone = _zone_bounds(src, geozone_contour_geometries)
        rb = src.bounds
        crs = getattr(src, "crs", None)
    finally:
        if isinstance(img_path, (str, bytes)):
            src.close()
    if zone is None:
        return pd.DataFrame(columns=COLUMNS)

    tiles = slice_bounds(zone, (rb.left, rb.bottom, rb.right, rb.top), patch_size, margin, resolution)
    input_id = img_path if isinstance(img_path, (str, bytes)) else getattr(img_path, "name", "<raster>")
    rows = [{
        "id": f"1-{t.row}-{t.col}", "input_id": input_id, "output_id": output_name, "job_done": 0,
        "left": t.left, "bottom": t.bottom, "right": t.right, "top": t.top,
        "left_o": zone[0], "bottom_o": zone[1], "right_o": zone[2], "top_o": zone[3],
        "geometry": create_box_from_bounds(t.x0, t.x1, t.y0, t.y1),
    } for t in tiles]
    try:
        import geopandas as gpd  # type: ignore
        out = gpd.GeoDataFrame(rows, crs=crs, geometry="geometry") if rows else gpd.GeoDataFrame()
    except ImportError:
        out = pd.DataFrame(rows, columns=COLUMNS)
    if config.get("write_dataframe", False) and hasattr(out, "to_file"):
        path = os.path.join(config["output_path"], output_name + "_slicing_job.gpkg")
        out.to_file(path, driver="GPKG")
        logger.info("saved sliced boxes: %s", path)
    return out
