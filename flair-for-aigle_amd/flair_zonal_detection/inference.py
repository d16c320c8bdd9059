"""Zonal tiled inference -- counterpart of the reference's flair_zonal_detection/inference.py
(prep_config :54-73, initialize_geometry_and_resolutions :76-132, prep_dataset :136-154, init_outputs
:157-208, inference_and_write :254-355, run_inference :644-674).

The tile loop keeps the reference's semantics -- per tile: drop the margin, convert to uint8, place the
window at int(round((left - L) / res)), int(round((T - top) / res)), clip to the raster, skip empty windows,
last writer wins -- and moves the per-pixel work onto the GPU:

    reference (per tile)                             here (per batch)
    D2H of f32 logits, 19.9 MB                       margin crop + argmax / class_prob fused in one HIP kernel
    numpy argmax over 19 x 432 x 432                 over the whole batch (ffa_predict_u8), D2H of uint8:
    scipy nearest zoom (optional)                    0.19 MB per tile; window maths by ffa_write_window
    rasterio window write                            (bit-exact with inference.py:318-335), then the same write

Polygonisation (:359-466, :566-630), geozone loading (:229-252) and COG conversion are product glue outside
the hot path and are not provided.
"""
from __future__ import annotations

import glob
import logging
import os
import time
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader

from flairhip import ops
from flair_zonal_detection.config import config_recap_1, config_recap_2, load_config, validate_config
from flair_zonal_detection.dataset import MultiModalSlicedDataset, TileBatcher, pad_series_collate
from flair_zonal_detection.model_utils import build_inference_model, compute_patch_sizes
from flair_zonal_detection.postprocess import convert  # noqa: F401  (re-exported like the reference)
from flair_zonal_detection.raster import ArrayRaster, make_window, open_raster
from flair_zonal_detection.slicing import generate_patches_from_reference

logger = logging.getLogger(__name__)


def overwrite_config(config: dict, model_ckpt_path: str, model_threshold_filepath: str, result_folder: str,
                     log_folder: str) -> Dict:
    config["model_weights"] = model_ckpt_path
    config["model_threshold_filepath"] = model_threshold_filepath
    config["output_path"] = result_folder
    config["log_folder"] = log_folder
    return config


def prep_config(config_path: str, model_ckpt_path: Optional[str] = None, model_threshold_filepath: Optional[str] = None,
                result_folder: Optional[str] = None, log_folder: Optional[str] = None,
                images_folder: Optional[str] = None) -> Dict:
    """Load + validate the YAML, resolve geometry, pick the device.  Accepts the fork's six-argument call
    (scripts/run_fast_aigle_segmentation.py:75) and upstream FLAIR-HUB's one-argument call."""
    config = load_config(config_path) if isinstance(config_path, str) else config_path
    if images_folder is not None:
        # the fork's caller feeds BD ORTHO JPEG-2000 tiles (scripts/run_fast_aigle_segmentation.py:88): opened by
        # rasterio (GDAL) when installed, else by jp2.Jp2Raster (OpenJPEG through Pillow); with neither only GeoTIFF
        # rasters can be opened (geotiff.py), so say so HERE instead of globbing *.jp2 and failing at the first read
        from flair_zonal_detection import jp2
        try:
            import rasterio  # type: ignore  # noqa: F401
            can_jp2 = True
        except ImportError:
            can_jp2 = jp2.openjpeg_available()
        patterns = ("*.jp2", "*.tif", "*.tiff") if can_jp2 else ("*.tif", "*.tiff")
        if not can_jp2 and glob.glob(os.path.join(images_folder, "*.jp2")):
            logger.warning("%s holds JPEG-2000 rasters, which need rasterio / GDAL or a Pillow with OpenJPEG (neither "
                           "installed): only GeoTIFF inputs are considered", images_folder)
        rasters = sorted(p for pat in patterns for p in glob.glob(os.path.join(images_folder, pat)))
        if not rasters:
            raise FileNotFoundError(f"no raster ({', '.join(patterns)}) in {images_folder}"
                                    + ("" if can_jp2 else
                                       "; JPEG-2000 inputs need rasterio or a Pillow with OpenJPEG, which are not installed"))
        config["modalities"]["AERIAL_RGBI"]["input_img_path"] = rasters[0]
    if model_ckpt_path is not None:
        config = overwrite_config(config, model_ckpt_path, model_threshold_filepath, result_folder, log_folder)
    validate_config(config)
    config_recap_1(config)
    config = initialize_geometry_and_resolutions(config)
    config_recap_2(config)
    use_gpu = config.get("use_gpu", True)
    if not (use_gpu and torch.cuda.is_available()):
        raise RuntimeError("the libflairhip tile loop needs an MI355X (use_gpu must be true and a GPU present); "
                           "there is no CPU path in the product")
    config["device"] = torch.device("cuda")
    config["output_type"] = config.get("output_type", "argmax")
    return config


def initialize_geometry_and_resolutions(config: Dict) -> Dict:
    """reference_resolution (finest active modality, rounded to 5 decimals), per-modality resolutions, image
    bounds / shape, tile and margin size in metres; bounds of all active modalities must agree within 1e-2."""
    modalities = config["modalities"]
    active = [m for m, on in modalities["inputs"].items() if on]
    resolutions, bounds = {}, []
    for mod in active:
        path = modalities[mod]["input_img_path"]
        src = open_raster(path)
        try:
            resolutions[mod] = round(src.res[0], 5)
            bounds.append((mod, src.bounds))
            if "image_shape_px" not in config:
                config["image_shape_px"] = {"height": src.height, "width": src.width}
        finally:
            if isinstance(path, (str, bytes)):
                src.close()
    ref_mod, ref_bounds = bounds[0]
    for mod, b in bounds[1:]:
        if not np.allclose(tuple(b), tuple(ref_bounds), atol=1e-2):
            raise ValueError(f"Bounds mismatch between '{ref_mod}' and '{mod}':\n  {ref_mod}: {ref_bounds}\n  {mod}: {b}")
    ref_mod, reference_resolution = min(resolutions.items(), key=lambda kv: kv[1])
    config["reference_modality"] = ref_mod
    config["reference_resolution"] = reference_resolution
    config["modality_resolutions"] = resolutions
    config["image_bounds"] = {"left": ref_bounds.left, "bottom": ref_bounds.bottom, "right": ref_bounds.right,
                              "top": ref_bounds.top}
    config["tile_size_m"] = round(config["img_pixels_detection"] * reference_resolution, 2)
    config["margin_size_m"] = round(config["margin"] * reference_resolution, 2)
    return config


def prep_dataset(config: Dict, tiles_gdf, patch_sizes: Dict[str, int]) -> MultiModalSlicedDataset:
    active = [m for m, on in config["modalities"]["inputs"].items() if on]
    config["labels"] = [t["name"] for t in config["tasks"] if t["active"]]
    config["labels_configs"] = {t["name"]: {"value_name": t["class_names"]} for t in config["tasks"] if t["active"]}
    return MultiModalSlicedDataset(dataframe=tiles_gdf, modality_cfgs={m: config["modalities"][m] for m in active},
                                   patch_size_dict=patch_sizes, ref_date_str=config.get("multitemp_model_ref_date", "05-15"),
                                   modalities_config=config,
                                   device_normalize=bool(config.get("device_normalize", True)))


def init_outputs(config: Dict, ref_img, i=None) -> Tuple[Dict[str, object], Dict[str, str]]:
    """One uint8 output raster per active task (1 band for argmax, K for class_prob), on the reference raster's
    grid or on an output_px_meters grid when that differs from the reference resolution."""
    output_type = config["output_type"]
    ref_res = config["reference_resolution"]
    out_res = config.get("output_px_meters", ref_res)
    ib = config["image_bounds"]
    needs_rescale = abs(ref_res - out_res) > 1e-6
    outputs, paths = {}, {}
    for task in config["tasks"]:
        if not task["active"]:
            continue
        k = len(task["class_names"])
        count = k if output_type == "class_prob" else 1
        suffix = "argmax" if output_type == "argmax" else "class-prob"
        path = os.path.join(config["output_path"], f"{config['output_name']}_{task['name']}_{suffix}_i.tif")
        if config.get("shard") is not None:  # one part file per rank; geotiff.merge_shard_files joins them
            path = path[:-4] + ".r{}of{}.tif".format(*config["shard"])
        if needs_rescale:
            h = int(round((ib["top"] - ib["bottom"]) / out_res))
            w = int(round((ib["right"] - ib["left"]) / out_res))
        else:
            h, w = ref_img.height, ref_img.width
        if isinstance(ref_img, ArrayRaster):
            outputs[task["name"]] = ArrayRaster(np.zeros((count, h, w), np.uint8), ib["left"], ib["top"], out_res,
                                                ref_img.crs)
        else:
            try:
                import rasterio  # type: ignore
            except ImportError:
                rasterio = None
            # a sharded run joins per-rank part files + "written" masks (geotiff.merge_shard_files): that protocol
            # belongs to the built-in writer, so it is used for the parts even where rasterio is installed
            if rasterio is None or config.get("shard") is not None:  # tiled, LZW, written on close()
                from flair_zonal_detection.geotiff import GeoTiffWriter
                os.makedirs(config["output_path"], exist_ok=True)
                outputs[task["name"]] = GeoTiffWriter.like(path, ref_img, count, np.uint8, width=w, height=h,
                                                           left=ib["left"], top=ib["top"],
                                                           res=out_res if needs_rescale else ref_img.res)
            else:
                from rasterio.transform import from_origin  # type: ignore
                profile = ref_img.profile.copy()
                profile.update({"count": count, "dtype": "uint8", "compress": "lzw"})
                if needs_rescale:
                    profile.update({"driver": "GTiff", "height": h, "width": w,
                                    "transform": from_origin(ib["left"], ib["top"], out_res, out_res)})
                outputs[task["name"]] = rasterio.open(path, "w", **profile)
        paths[task["name"]] = path
    return outputs, paths


def _zoom_index(n: int, scale: float):
    """Source index of every output position of ``scipy.ndimage.zoom(x, scale, order=0)`` along an axis of length
    n, or -1 where scipy writes its constant 0 (reference inference.py:212-226 calls zoom with the default
    mode='constant', grid_mode=False).  scipy maps output i to the input coordinate cc = i * (n - 1) / (out - 1),
    out = round(n * scale), in float64, takes floor(cc + 0.5) for order 0 and treats cc > n - 1 -- which float
    rounding produces for the LAST position of some (n, scale) pairs, e.g. n = 226, scale 0.5 -- as outside the
    array.  Checked against scipy on 1.8 M positions (tests/test_oracle_goldens.py)."""
    import math
    out = int(round(n * scale))
    step = (n - 1) / (out - 1) if out > 1 else 1.0
    idx = []
    for i in range(out):
        cc = i * step
        idx.append(-1 if (cc < 0.0 or cc > n - 1) else int(math.floor(cc + 0.5)))
    return idx


def _nearest_zoom(pred: torch.Tensor, scale: float) -> torch.Tensor:
    """scipy.ndimage.zoom(order=0) on the last two axes, position for position (resample_prediction, reference
    inference.py:212-226)."""
    h, w = pred.shape[-2:]
    ys = torch.tensor(_zoom_index(h, scale), dtype=torch.long, device=pred.device)
    xs = torch.tensor(_zoom_index(w, scale), dtype=torch.long, device=pred.device)
    out = pred[..., ys.clamp(min=0), :][..., xs.clamp(min=0)]
    if bool((ys < 0).any()) or bool((xs < 0).any()):
        out = out.clone()
        out[..., ys < 0, :] = 0
        out[..., xs < 0] = 0
    return out


@torch.no_grad()
def inference_and_write(model: torch.nn.Module, dataloader: DataLoader, tiles_gdf, config: Dict,
                        output_files: Dict[str, object], ref_img) -> None:
    device = config["device"]
    margin = config["margin"]
    tile_size = config["img_pixels_detection"]
    output_type = config["output_type"]
    if output_type not in ("argmax", "class_prob"):
        raise ValueError(f"Unknown output type: {output_type}")
    ref_res = config["reference_resolution"]
    out_res = config.get("output_px_meters", ref_res)
    needs_rescale = abs(ref_res - out_res) > 1e-6
    scale = ref_res / out_res if needs_rescale else 1.0
    img_bounds = tuple(ref_img.bounds)  # (left, bottom, right, top)
    keep = tile_size - 2 * margin

    # raw raster tiles (dataset with device_normalize): the per-channel (mean, std) ride along once per modality
    norms = {}
    ds = getattr(dataloader, "dataset", None)
    if getattr(ds, "device_normalize", False):
        for mod in ds.modalities:
            if ds.delivers_raw(mod):
                norms[mod + "_NORM"] = torch.tensor(np.stack(ds.norm_vectors(mod)), dtype=torch.float32, device=device)

    # tile columns as plain arrays: a pandas row lookup costs ~0.5 ms, twice per tile
    lefts, tops, ids = (np.asarray(tiles_gdf[c]) for c in ("left", "top", "id"))
    graphed = None  # hipGraph of forward + conversion for full batches (the eval forward is ~120 launches)
    use_graph = bool(config.get("hip_graph", True)) and str(device).startswith("cuda")

    def forward_eager(inputs):
        logits_tasks, _ = model(inputs)
        preds = {}
        for task_name, logits in logits_tasks.items():
            preds[task_name] = ops.predict_u8(logits._ffa_nhwc, logits._ffa_classes, output_type,
                                              crop=(margin, margin, keep, keep))
        return preds

    def write_batch(indices, host_preds):
        """host side of one batch: window placement + raster writes (reference inference.py:297-352)"""
        for task_name, (buf, done) in host_preds.items():
            done.synchronize()
            pred = buf[:len(indices)].numpy()  # uint8: [B,h,w] or [B,K,h,w]
            for i in range(len(indices)):
                ti = int(indices[i])
                p = pred[i]
                win = ops.write_window(lefts[ti], tops[ti], img_bounds, out_res, p.shape[-2], p.shape[-1])
                if win.skip:
                    logger.info("skipping tile %s: window out of bounds", ids[ti])
                    continue
                p = p[..., :win.height, :win.width]
                window = make_window(win.col_off, win.row_off, win.width, win.height)
                if output_type == "argmax":
                    output_files[task_name].write(p, 1, window=window)
                else:
                    for c in range(p.shape[0]):
                        output_files[task_name].write(p[c], c + 1, window=window)

    # Two-stage software pipeline: the device works on batch k (H2D, forward, conversion, D2H into a pinned buffer,
    # all stream-ordered and asynchronous) while the host writes the windows of batch k-1 and reads the tiles of
    # batch k+1.  Pinned D2H buffers alternate; the loader's input buffers do too (TileBatcher).
    full = getattr(dataloader, "bs", None) or getattr(dataloader, "batch_size", None)
    host_bufs: Dict[str, list] = {}
    pending = None
    for k, batch in enumerate(dataloader):
        inputs = {k_: v.to(device, non_blocking=True) for k_, v in batch.items()
                  if k_ != "index" and torch.is_tensor(v)}
        for k_, v in norms.items():
            if inputs.get(k_[:-5]) is not None:
                inputs[k_] = v
        indices = batch["index"].cpu().numpy().flatten()
        if use_graph and full and len(indices) == full:
            if graphed is None:
                from flairhip.graph import GraphedCall
                graphed = GraphedCall(forward_eager, inputs)
            preds = graphed(inputs)
        else:
            preds = forward_eager(inputs)
        host_preds = {}
        for task_name, pred in preds.items():
            if needs_rescale:
                pred = _nearest_zoom(pred, scale)
            bufs = host_bufs.get(task_name)
            if bufs is None or bufs[0].shape[1:] != pred.shape[1:] or bufs[0].shape[0] < pred.shape[0]:
                shape = (max(int(full or 0), pred.shape[0]),) + tuple(pred.shape[1:])
                bufs = host_bufs[task_name] = [torch.empty(shape, dtype=pred.dtype).pin_memory() for _ in range(2)]
            buf = bufs[k & 1]
            buf[:pred.shape[0]].copy_(pred, non_blocking=True)  # before the next replay overwrites the graph's output
            done = torch.cuda.Event()
            done.record()
            host_preds[task_name] = (buf, done)
        if pending is not None:
            write_batch(*pending)
        pending = (indices, host_preds)
    if pending is not None:
        write_batch(*pending)
    for dst in output_files.values():
        dst.close()


def shard_tiles(tiles, rank: int, world: int):
    """Tiles of one rank of a sharded run: a CONTIGUOUS slice of the grid order (outer loop x, inner y).  No
    collective is needed -- tiles are independent units -- and contiguity keeps the reference's "last writer wins"
    for the clamped last row / column inside a shard; across shards merge_shard_outputs restores it by rank order."""
    if not (0 <= rank < world):
        raise ValueError(f"shard rank {rank} outside world size {world}")
    n = len(tiles)
    return tiles.iloc[rank * n // world:(rank + 1) * n // world].reset_index(drop=True)


def merge_shard_outputs(outputs_by_rank):
    """One set of output rasters from the per-rank rasters of a sharded in-memory run (ArrayRaster with
    track_writes): pixels a later rank wrote replace those of earlier ranks, i.e. grid order = write order."""
    merged = {}
    for task, first in outputs_by_rank[0].items():
        out = ArrayRaster(first.data.copy(), first.left, first.top, first._res, first.crs)
        for part in outputs_by_rank[1:]:
            r = part[task]
            if r.written is None:
                raise ValueError("merge_shard_outputs needs rasters that tracked their writes")
            out.data[:, r.written] = r.data[:, r.written]
        merged[task] = out
    return merged


def run_inference(config_path, ref_raster=None, geozone=None, shard: Optional[Tuple[int, int]] = None,
                  before_loop=None) -> Dict[str, object]:
    """End-to-end zonal run with upstream FLAIR-HUB's one-argument semantics (the fork's own run_inference is
    stale: inference.py:650-665 calls its helpers with the wrong arity).  Returns the output rasters.
    ``shard=(rank, world)`` (or config['shard']) restricts the run to that rank's slice of the tile grid: one process
    per GPU, no communication; in-memory outputs then track their writes for merge_shard_outputs.  ``before_loop``
    (optional) is called with the freshly initialised outputs before the first tile is processed."""
    t0 = time.time()
    config = prep_config(config_path)
    ref_path = config["modalities"][config["reference_modality"]]["input_img_path"]
    ref_img = ref_raster if ref_raster is not None else open_raster(ref_path)
    tiles = generate_patches_from_reference(config, ref_img, geozone)
    shard = shard if shard is not None else config.get("shard")
    if shard is not None:
        config["shard"] = (int(shard[0]), int(shard[1]))
        tiles = shard_tiles(tiles, int(shard[0]), int(shard[1]))
    patch_sizes = compute_patch_sizes(config)
    model = build_inference_model(config, patch_sizes).to(config["device"])
    dataset = prep_dataset(config, tiles, patch_sizes)
    if TileBatcher.supports(dataset) and not config.get("num_worker", 0):
        loader = TileBatcher(dataset, config.get("batch_size", 8))  # uint8 tiles straight into pinned batch buffers
    else:
        series = any(m.endswith("_TS") for m in dataset.modalities)
        loader = DataLoader(dataset, batch_size=config.get("batch_size", 8), num_workers=config.get("num_worker", 0),
                            pin_memory=True, collate_fn=pad_series_collate if series else None)
        if series and dataset.mask_reader is not None:
            config["hip_graph"] = False  # per-tile cloud filtering: the number of dates changes from batch to batch
    outputs, _ = init_outputs(config, ref_img)
    if before_loop is not None:
        before_loop(outputs)
    if shard is not None:
        for o in outputs.values():
            if isinstance(o, ArrayRaster):
                o.track_writes()
    inference_and_write(model, loader, tiles, config, outputs, ref_img)
    logger.info("zonal inference of %d tiles took %.1f s", len(tiles), time.time() - t0)
    return outputs
