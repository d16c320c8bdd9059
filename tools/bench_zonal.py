#!/usr/bin/env python
"""Zonal inference throughput on a synthetic in-memory raster.

  python tools/bench_zonal.py [--size 6048] [--batch 8]

Reports (a) the model forward alone (eval mode: BatchNorm folded into the conv operands, bias / ReLU / decoder
upsample+concat in conv epilogues / prologues) at the loop's batch size and at 32, and (b) the whole
run_inference loop: slicing, windowed reads + normalisation (numpy, host), H2D, forward, fused margin-crop + argmax,
D2H of 1 byte per kept pixel, window placement, writes into the in-memory output raster.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))

import numpy as np
import torch
import yaml


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=6048, help="raster height = width in pixels")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--output-type", default="argmax", choices=["argmax", "class_prob"])
    ap.add_argument("--profile", action="store_true", help="cProfile of the tile loop (host hot spots)")
    ap.add_argument("--tif", action="store_true", help="also run from / to GeoTIFF files (built-in reader / writer)")
    ap.add_argument("--arch", default="resnet34-unet",
                    help="models.monotemp_model.arch, e.g. swin_base_patch4_window12_384-upernet (the fork's zonal config)")
    ap.add_argument("--channels", type=int, default=5)
    ap.add_argument("--forward-only", action="store_true", help="skip the tile loop (kernel profiles of the network)")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    from flairhip.configs import unet_resnet34_config
    from flair_hub.models.flair_model import FLAIR_HUB_Model
    from flair_zonal_detection.inference import run_inference
    from flair_zonal_detection.raster import ArrayRaster

    dev = torch.device("cuda:0")
    MOD, TASK = "AERIAL_RGBI", "AERIAL_LABEL-COSIA"
    # (a) forward alone
    C = args.channels
    cfg = unet_resnet34_config(in_channels=C, precision=args.precision)
    cfg["models"]["monotemp_model"]["arch"] = args.arch
    model = FLAIR_HUB_Model(cfg, {MOD: 512}).to(dev).eval()
    for B in ((args.batch,) if args.forward_only else (args.batch, 32)):
        x = torch.randn(B, C, 512, 512, device=dev)
        with torch.no_grad():
            for _ in range(3):
                model({MOD: x})
            torch.cuda.synchronize()
            t0 = time.time()
            n = args.iters
            for _ in range(n):
                model({MOD: x})
            torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        print(f"forward only, batch {B:2d}: {dt * 1e3:7.2f} ms/batch = {B / dt:8.1f} tiles/s")

    if args.forward_only:
        return
    # (b) the zonal loop
    g = np.random.default_rng(0)
    H = W = args.size
    img = g.integers(0, 255, (C, H, W), dtype=np.uint8)
    ras = ArrayRaster(img, 651000.0, 6865000.0, 0.2)
    zc = yaml.safe_load(open(os.path.join(ROOT, "tests", "golden", "zonal_config.yaml")))
    zc.update({"output_path": "/tmp", "output_name": "bench_zonal", "img_pixels_detection": 512, "margin": 40,
               "output_px_meters": 0.2, "output_type": args.output_type, "batch_size": args.batch, "num_worker": 0,
               "hardware": {"precision": args.precision}, "model_weights": "/tmp/bench_zonal_weights.ckpt",
               "monotemp_arch": args.arch})
    torch.save({"state_dict": {"model." + k: v.cpu() for k, v in model.state_dict().items()}}, zc["model_weights"])
    zc["modalities"][MOD].update({"input_img_path": ras, "channels": list(range(1, C + 1)),
                                  "normalization": {"type": "custom", "means": [110.0] * C, "stds": [50.0] * C}})
    zc["tasks"] = [{"name": TASK, "active": True, "class_names": {i: f"c{i}" for i in range(19)}}]
    t0 = time.time()
    out = run_inference(zc)
    torch.cuda.synchronize()
    dt = time.time() - t0
    ntiles = ((H + 80 + 431) // 432) ** 2
    print(f"run_inference on {H}x{W} px ({ntiles} tiles of 512, batch {args.batch}): {dt:.2f} s = {ntiles / dt:.1f} tiles/s, "
          f"{H * W / dt / 1e6:.1f} Mpx/s; output {out[TASK].data.shape}")

    # the same run split into its stages (second pass: kernels and workspaces are warm)
    from torch.utils.data import DataLoader
    from flair_zonal_detection import inference as zi
    from flair_zonal_detection.model_utils import build_inference_model, compute_patch_sizes
    from flair_zonal_detection.slicing import generate_patches_from_reference
    t = [time.time()]
    cfg2 = zi.prep_config(zc)
    tiles = generate_patches_from_reference(cfg2, ras, None)
    t.append(time.time())
    sizes = compute_patch_sizes(cfg2)
    mdl = build_inference_model(cfg2, sizes).to(cfg2["device"])
    t.append(time.time())
    ds = zi.prep_dataset(cfg2, tiles, sizes)
    from flair_zonal_detection.dataset import TileBatcher
    loader = TileBatcher(ds, args.batch) if TileBatcher.supports(ds) else DataLoader(ds, batch_size=args.batch,
                                                                                     pin_memory=True)
    outputs, _ = zi.init_outputs(cfg2, ras)
    t.append(time.time())
    if args.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
    zi.inference_and_write(mdl, loader, tiles, cfg2, outputs, ras)
    torch.cuda.synchronize()
    t.append(time.time())
    if args.profile:
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    names = ["config + slicing", "model build + checkpoint", "dataset + output rasters", "tile loop"]
    print("  stages: " + ", ".join(f"{n} {b - a:.2f} s" for n, a, b in zip(names, t, t[1:])) +
          f"  -> tile loop alone {len(tiles) / (t[4] - t[3]):.0f} tiles/s")

    if args.tif:  # the same mosaic as an LZW GeoTIFF on disk, predictions written as GeoTIFF
        import copy
        import tempfile
        from flair_zonal_detection.geotiff import GeoTiffWriter
        d = tempfile.mkdtemp(prefix="bench_zonal_")
        src = os.path.join(d, "mosaic.tif")
        # label-like smooth content so that LZW has something to do (noise would not compress at all)
        smooth = np.repeat(np.repeat(g.integers(0, 255, (5, H // 8 + 1, W // 8 + 1), dtype=np.uint8), 8, 1), 8, 2)[:, :H, :W]
        for comp in ("lzw", None):
            t0 = time.time()
            with GeoTiffWriter.like(src, ras, 5, compress=comp) as w:
                w.data[...] = smooth
            t1 = time.time()
            zt = copy.deepcopy(zc)
            zt["modalities"][MOD]["input_img_path"] = src
            zt["output_path"] = d
            out = run_inference(zt)
            torch.cuda.synchronize()
            dt = time.time() - t1
            print(f"GeoTIFF in ({comp or 'uncompressed'}, {os.path.getsize(src) / 1e6:.0f} MB, written in {t1 - t0:.1f} s) "
                  f"-> GeoTIFF out ({os.path.getsize(out[TASK].path) / 1e6:.1f} MB): {dt:.2f} s = {ntiles / dt:.1f} tiles/s")
        import shutil
        shutil.rmtree(d)


if __name__ == "__main__":
    main()
