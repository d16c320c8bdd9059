"""Tile dataset of the zonal loop -- counterpart of the reference's flair_zonal_detection/dataset.py
(MultiModalSlicedDataset :24, _load_patch :89-117, _process_time_series_patch :121-169, __getitem__ :174-209).

Per tile and modality: a boundless windowed read of the tile box (zero fill outside the raster, bilinear
resample to the modality's patch size) followed by the per-channel (x - mean) / std normalisation of
flair_hub/data/utils_data/norm.py:37-44 ('custom'); 'without' leaves values as they are.

Time-series modalities ('<SENSOR>_TS', :100-104, :121-169): all T x C bands of the stack are read for the tile box, reshaped
to [T, C, h, w] (no normalisation, as in the reference), optionally filtered by the Sentinel-2 cloud / snow mask raster
(nearest-neighbour read, filter_time_series) and / or averaged per month or half-month (temporal_average); the day offsets
to the reference date ride along as '<SENSOR>_DATES'.

Differences from the reference, all on the host side of the boundary:
  * cloud filtering is decided PER TILE and does not touch the dataset's date table: the reference overwrites
    self.diff_dates with the first tile's surviving dates (dataset.py:155-156), after which every later tile reads the
    wrong number of bands; tiles of one batch may then hold different numbers of dates -- collate with
    ``pad_series_collate`` (zero-padded like the training pipeline's pad_collate_flair: all-zero dates are what the
    U-TAE treats as padding)
  * the raster objects are duck-typed (flair_zonal_detection.raster) instead of rasterio-only
  * the '<MOD>_RAW' copy and the zero '<TASK>' label of the reference (dataset.py:194-207; 3 MB + 19.9 MB
    of H2D traffic per tile that the model never reads) are only emitted with ``reference_batch_schema=True``;
    FLAIR_HUB_Model falls back to the input size when no label rides along.
"""
from __future__ import annotations

from typing import Any, Dict

import numpy as np
import torch
from torch.utils.data import Dataset

from flair_zonal_detection.raster import open_raster


def normalize_array(img: np.ndarray, norm_type, means, stds) -> np.ndarray:
    if norm_type not in ("scaling", "custom", "without", None):
        raise ValueError("Normalization argument should be 'scaling', 'custom', or 'without'.")
    if norm_type == "custom":
        if len(means) != len(stds):
            raise ValueError("If using 'custom', the provided means and stds must have the same length.")
        img = img.astype(np.float64)
        for i in range(img.shape[0]):
            img[i] -= means[i]
            img[i] /= stds[i]
    elif norm_type == "scaling":
        info = np.iinfo(img.dtype) if np.issubdtype(img.dtype, np.integer) else None
        img = img.astype(np.float64) / info.max if info is not None else img.astype(np.float64)
    return img


# sample types the layout kernel normalises on the device (ffa_u8_nchw_to_nhwc / ffa_raw_nchw_to_nhwc)
RAW_DTYPES = {np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.uint16, np.dtype(np.int16): torch.int16,
              np.dtype(np.float32): torch.float32}


class MultiModalSlicedDataset(Dataset):
    def __init__(self, dataframe, modality_cfgs: Dict[str, Dict[str, Any]], patch_size_dict: Dict[str, int],
                 ref_date_str: str, modalities_config: Dict[str, Any], reference_batch_schema: bool = False,
                 device_normalize: bool = False) -> None:
        self.df = dataframe
        self.modalities = modality_cfgs
        self.modalities_config = modalities_config
        self.patch_sizes = patch_size_dict
        self.ref_date_str = ref_date_str
        self.reference_batch_schema = reference_batch_schema
        # uint8 rasters with 'custom' / 'scaling' normalisation: hand the raw bytes over and let the layout kernel on
        # the device normalise (ffa_u8_nchw_to_nhwc): a quarter of the PCIe bytes, no per-tile float work here
        self.device_normalize = device_normalize and not reference_batch_schema
        self.readers = {m: open_raster(cfg["input_img_path"]) for m, cfg in modality_cfgs.items()}
        # Sentinel-2 cloud / snow mask stack (2 bands per date), possibly at another resolution (:51-56)
        self.mask_reader, self.mask_resolution_ratio = None, 1.0
        s2 = modality_cfgs.get("SENTINEL2_TS")
        if s2 and s2.get("filter_clouds") and "filter_clouds_img_path" in s2:
            self.mask_reader = open_raster(s2["filter_clouds_img_path"])
            self.mask_resolution_ratio = self.readers["SENTINEL2_TS"].res[0] / self.mask_reader.res[0]
        self.series_dates = self._init_series_dates()

    def _init_series_dates(self):
        """per time-series modality: acquisition dates (dates_txt, one YYYYMMDD per line) and their day offsets to the
        reference month-day of the same year (:60-87)"""
        from datetime import datetime
        ref_month, ref_day = (int(v) for v in self.ref_date_str.split("-"))
        out = {}
        for mod, cfg in self.modalities.items():
            if not mod.endswith("_TS"):
                continue
            if not cfg.get("dates_txt"):
                raise ValueError(f"'dates_txt' is required for the time-series modality '{mod}' (one YYYYMMDD per line, "
                                 "in the band order of the stack)")
            with open(cfg["dates_txt"]) as f:
                strs = [line.strip() for line in f if line.strip()]
            if not strs:
                raise ValueError(f"'dates_txt' file for '{mod}' is empty.")
            dates = [datetime.strptime(d, "%Y%m%d") for d in strs]
            out[mod] = {"dates": dates,
                        "diff_dates": np.array([(d - datetime(d.year, ref_month, ref_day)).days for d in dates])}
        return out

    def _load_series(self, mod: str, bounds, cfg, patch_size: int):
        """-> (float32 [T', C, h, w], float32 [T'] day offsets) of one tile"""
        from flair_hub.data.utils_data.sentinel import filter_time_series, reshape_sentinel, temporal_average
        info = self.series_dates[mod]
        dates, diffs = list(info["dates"]), np.asarray(info["diff_dates"])
        C = len(cfg["channels"])
        reader = self.readers[mod]
        if reader.count < C * len(dates):
            raise ValueError(f"'{mod}' has {reader.count} bands, {len(dates)} dates x {C} channels were announced")
        patch = reader.read_bounds(list(range(1, C * len(dates) + 1)), bounds, patch_size)
        patch = reshape_sentinel(np.asarray(patch, dtype=np.float32), C)
        if mod == "SENTINEL2_TS" and self.mask_reader is not None:
            hm = int(patch.shape[2] / self.mask_resolution_ratio)
            msk = self.mask_reader.read_bounds(list(range(1, 2 * len(dates) + 1)), bounds, max(hm, 1), nearest=True)
            keep = filter_time_series(reshape_sentinel(np.asarray(msk), 2))
            if keep.sum() > 0:
                patch, diffs = patch[keep], diffs[keep]
                dates = [d for d, k in zip(dates, keep) if k]
        if cfg.get("temporal_average", False):
            patch, diffs = temporal_average(patch, dates, period=cfg.get("average_period", "monthly"),
                                            ref_date=self.ref_date_str)
        return np.ascontiguousarray(patch, dtype=np.float32), np.ascontiguousarray(diffs, dtype=np.float32)

    def __len__(self) -> int:
        return len(self.df)

    @staticmethod
    def _tile_box(row):
        g = row["geometry"]
        return g.bounds if hasattr(g, "bounds") else tuple(g)  # (minx, miny, maxx, maxy)

    def _load_patch(self, reader, bounds, cfg, patch_size: int) -> np.ndarray:
        if hasattr(reader, "read_bounds"):
            return reader.read_bounds(cfg["channels"], bounds, patch_size)
        from rasterio.enums import Resampling  # type: ignore
        from rasterio.windows import from_bounds  # type: ignore
        window = from_bounds(*bounds, transform=reader.transform)
        return reader.read(indexes=cfg["channels"], window=window, out_shape=(len(cfg["channels"]), patch_size,
                           patch_size), resampling=Resampling.bilinear, boundless=True, fill_value=0)

    def delivers_raw(self, mod: str) -> bool:
        """True when tiles of ``mod`` leave this dataset as raw raster samples (+ '<MOD>_NORM' rides along in the loop)"""
        if not self.device_normalize or self.norm_vectors(mod) is None:
            return False
        dt = getattr(self.readers[mod], "dtype", None)
        if dt is None and hasattr(self.readers[mod], "dtypes"):  # rasterio dataset
            dt = self.readers[mod].dtypes[0]
        try:
            return np.dtype(dt).newbyteorder("=") in RAW_DTYPES
        except TypeError:
            return False

    def norm_vectors(self, mod: str):
        """(mean, std) per channel of the normalisation the device applies to uint8 tiles of ``mod``, or None"""
        cfg = self.modalities[mod]
        ncfg = cfg.get("normalization") or {}
        n = len(cfg["channels"])
        if ncfg.get("type") == "custom":
            if len(ncfg["means"]) != len(ncfg["stds"]):
                raise ValueError("If using 'custom', the provided means and stds must have the same length.")
            return np.asarray(ncfg["means"][:n], np.float32), np.asarray(ncfg["stds"][:n], np.float32)
        if ncfg.get("type") == "scaling":
            return np.zeros(n, np.float32), np.full(n, 255.0, np.float32)
        return None

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        row = self.df.iloc[idx]
        bounds = self._tile_box(row)
        out: Dict[str, torch.Tensor] = {}
        for mod, cfg in self.modalities.items():
            if mod.endswith("_TS"):
                series, diffs = self._load_series(mod, bounds, cfg, self.patch_sizes[mod])
                out[mod] = torch.from_numpy(series)
                out[mod.replace("_TS", "_DATES")] = torch.from_numpy(diffs)
                continue
            patch = self._load_patch(self.readers[mod], bounds, cfg, self.patch_sizes[mod])
            ncfg = cfg.get("normalization", {})
            if self.device_normalize and patch.dtype in RAW_DTYPES and self.norm_vectors(mod) is not None:
                out[mod] = torch.from_numpy(np.ascontiguousarray(patch))  # raw samples: the device normalises
                continue
            norm = normalize_array(patch, ncfg.get("type"), ncfg.get("means"), ncfg.get("stds")) if ncfg else patch
            out[mod] = torch.tensor(np.ascontiguousarray(norm), dtype=torch.float32)
            if self.reference_batch_schema:
                out[mod + "_RAW"] = torch.tensor(np.ascontiguousarray(patch), dtype=torch.float32)
        out["index"] = torch.tensor([idx], dtype=torch.long)
        if self.reference_batch_schema:
            ref_size = list(self.patch_sizes.values())[0]
            for task in self.modalities_config["labels"]:
                k = len(self.modalities_config["labels_configs"][task]["value_name"])
                out[task] = torch.zeros((k, ref_size, ref_size), dtype=torch.float32)
        return out


def pad_series_collate(samples):
    """default_collate for zonal tiles whose time series may differ in length (per-tile cloud filtering): series and
    their date vectors are zero-padded at the end to the batch's longest (all-zero dates = padded dates for the U-TAE,
    the convention of flair_hub.data.utils_data.padding.pad_collate_flair)"""
    from torch.utils.data import default_collate
    keys = samples[0].keys()
    out = {}
    for k in keys:
        vals = [s[k] for s in samples]
        if torch.is_tensor(vals[0]) and (k.endswith("_TS") or k.endswith("_DATES")) and len({v.shape[0] for v in vals}) > 1:
            T = max(v.shape[0] for v in vals)
            vals = [torch.cat([v, v.new_zeros((T - v.shape[0],) + tuple(v.shape[1:]))]) if v.shape[0] < T else v
                    for v in vals]
        out[k] = default_collate(vals)
    return out


class TileBatcher:
    """Batches of raw raster tiles (uint8, or uint16 / int16 / float32) for the zonal loop, written by the raster reader
    straight into reused pinned host
    buffers: no per-tile tensor, no collate copy, no pin copy (torch's default DataLoader path spent 60 % of the
    loop's wall time in torch.stack and pandas row lookups).  Needs a dataset in device_normalize mode whose
    rasters read uint8 and accept ``read_bounds(..., out=)``; use ``TileBatcher.supports(dataset)``.

    Yields the same dict a DataLoader over the dataset would: {'<MOD>': uint8 [n,C,P,P], 'index': int64 [n,1]}.
    Four buffers per modality rotate, so a batch stays valid until three more have been requested (the reader thread
    runs one batch ahead of the consumer, which itself keeps one batch in flight on the device)."""

    NBUF = 4  # a buffer is rewritten four batches later: the loop has synchronised on that batch's output by then

    def __init__(self, dataset: MultiModalSlicedDataset, batch_size: int, prefetch: bool = True):
        if not self.supports(dataset):
            raise ValueError("TileBatcher needs device_normalize rasters of uint8 / uint16 / int16 / float32 samples "
                             "with read_bounds(..., out=)")
        self.ds = self.dataset = dataset
        self.bs = self.batch_size = int(batch_size)
        self.prefetch = bool(prefetch)  # read (and decode) the next batch on a worker thread while the device runs
        self.boxes = [dataset._tile_box(dataset.df.iloc[i]) for i in range(len(dataset))]
        self.bufs = {}
        for mod, cfg in dataset.modalities.items():
            shape = (self.bs, len(cfg["channels"]), dataset.patch_sizes[mod], dataset.patch_sizes[mod])
            tdt = RAW_DTYPES[np.dtype(dataset.readers[mod].dtype).newbyteorder("=")]
            self.bufs[mod] = [torch.empty(shape, dtype=tdt).pin_memory() if torch.cuda.is_available()
                              else torch.empty(shape, dtype=tdt) for _ in range(self.NBUF)]

    @staticmethod
    def supports(dataset) -> bool:
        if not getattr(dataset, "device_normalize", False):
            return False
        for mod, reader in dataset.readers.items():
            if dataset.norm_vectors(mod) is None or not hasattr(reader, "read_bounds"):
                return False
            if not dataset.delivers_raw(mod) or getattr(reader, "dtype", None) is None:  # ArrayRaster / GeoTiffRaster
                return False
        return True

    def __len__(self) -> int:
        return (len(self.ds) + self.bs - 1) // self.bs

    AHEAD = 4  # batches whose footprint is announced to the rasters (parallel block decode: JPEG-2000 mosaics)

    def _fill(self, k: int, start: int):
        idx = list(range(start, min(start + self.bs, len(self.ds))))
        out = {}
        if k % self.AHEAD == 0:  # union box of the next AHEAD batches: tiles run column by column, south to north
            ahead = self.boxes[start: start + self.AHEAD * self.bs]
            box = (min(b[0] for b in ahead), min(b[1] for b in ahead), max(b[2] for b in ahead), max(b[3] for b in ahead))
            for mod in self.ds.modalities:
                hint = getattr(self.ds.readers[mod], "prefetch_bounds", None)
                if hint is not None:
                    hint(box)
        for mod, cfg in self.ds.modalities.items():
            buf = self.bufs[mod][k % self.NBUF]
            arr = buf.numpy()
            for j, i in enumerate(idx):
                self.ds.readers[mod].read_bounds(cfg["channels"], self.boxes[i], self.ds.patch_sizes[mod], out=arr[j])
            out[mod] = buf[: len(idx)]
        out["index"] = torch.tensor(idx, dtype=torch.long).unsqueeze(1)
        return out

    def __iter__(self):
        starts = list(enumerate(range(0, len(self.ds), self.bs)))
        if not self.prefetch or len(starts) < 2:
            for k, start in starts:
                yield self._fill(k, start)
            return
        # one batch ahead on a worker thread: raster reads (block decode, big numpy copies) release the GIL
        import queue
        import threading
        q: "queue.Queue" = queue.Queue(maxsize=1)
        stop = threading.Event()

        def produce():
            try:
                for k, start in starts:
                    if stop.is_set():
                        return
                    item = self._fill(k, start)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                q.put(None)
            except BaseException as e:  # surfaces in the consumer
                q.put(e)

        th = threading.Thread(target=produce, name="tile-reader", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            th.join(timeout=5.0)
