"""JPEG-2000 input rasters (the fork's caller feeds BD ORTHO *.jp2,
scripts/run_fast_aigle_segmentation.py:88), CPU only: files are written by Pillow's OpenJPEG encoder (reversible 5/3
wavelet = lossless), the GeoJP2 box is a degenerate GeoTIFF from this repo's GeoTiffWriter spliced in front of the
code-stream box, and the reader's pixels / georeferencing / window reads are compared with the source array."""
import os
import struct

import numpy as np
import pytest

from flair_zonal_detection import jp2
from flair_zonal_detection.geotiff import GeoTiffError, GeoTiffRaster, GeoTiffWriter
from flair_zonal_detection.raster import ArrayRaster, make_window, open_raster

pytestmark = pytest.mark.skipif(not jp2.openjpeg_available(), reason="Pillow without OpenJPEG")


def geojp2_payload(tmp, left, top, res, crs):
    p = os.path.join(str(tmp), "_geo.tif")
    GeoTiffWriter(p, 1, 1, 1, left, top, res, crs=crs, compress=None).close()
    data = open(p, "rb").read()
    os.remove(p)
    return data


def write_jp2(path, arr, tmp=None, georef=None, xl=False):
    """arr [H, W] or [H, W, C] uint8 / [H, W] uint16 -> lossless JP2; georef = (left, top, res, crs) adds a GeoJP2 box
    (with a 64-bit XLBox length field when xl)"""
    from PIL import Image
    Image.fromarray(arr).save(path, format="JPEG2000", irreversible=False)
    if georef is None:
        return
    raw = open(path, "rb").read()
    body = jp2.GEOJP2_UUID + geojp2_payload(tmp, *georef)
    box = (struct.pack(">I4sQ", 1, b"uuid", 16 + len(body)) if xl else struct.pack(">I4s", 8 + len(body), b"uuid")) + body
    with open(path, "rb") as f:
        cut = next(off - 8 for t, off, n in jp2.iter_boxes(f) if t == b"jp2c")
    open(path, "wb").write(raw[:cut] + box + raw[cut:])


@pytest.mark.parametrize("shape,dtype,xl", [((70, 93), np.uint8, False), ((64, 80, 3), np.uint8, False),
                                           ((130, 257, 4), np.uint8, True), ((40, 33), np.uint16, False)])
def test_pixels_georeferencing_and_window_reads(tmp_path, shape, dtype, xl):
    g = np.random.default_rng(sum(shape))
    arr = g.integers(0, 255 if dtype == np.uint8 else 4000, shape).astype(dtype)
    p = str(tmp_path / "m.jp2")
    left, top, res = 651992.36, 6860417.84, 0.2
    write_jp2(p, arr, tmp_path, (left, top, res, "EPSG:2154"), xl=xl)
    assert jp2.is_jpeg2000(p)
    want = arr[None] if arr.ndim == 2 else arr.transpose(2, 0, 1)
    with open_raster(p) as r:
        assert isinstance(r, jp2.Jp2Raster) and r._data is None  # nothing decoded by opening
        assert (r.count, r.height, r.width) == want.shape and r.dtypes == (str(np.dtype(dtype)),) * want.shape[0]
        assert r.crs == "EPSG:2154" and r.res == (res, res)
        ref = ArrayRaster(want, left, top, res)
        assert np.allclose(tuple(r.bounds), tuple(ref.bounds), rtol=0, atol=1e-6)
        assert np.array_equal(r.read(), want)
        win = make_window(-5, 10, 40, 50)
        assert np.array_equal(r.read(window=win, boundless=True), ref.read(window=win, boundless=True))
        assert np.array_equal(r.read(1, window=make_window(3, 4, 9, 7)), want[0, 4:11, 3:12])
        b = r.bounds
        box = (b.left + 2 * res, b.top - 30 * res, b.left + 22 * res, b.top - 10 * res)
        idx = list(range(1, r.count + 1))
        assert np.array_equal(r.read_bounds(idx, box, 20), ref.read_bounds(idx, box, 20))
        assert np.array_equal(r.read_bounds(idx, box, 32), ref.read_bounds(idx, box, 32))  # resampled path
        assert r.profile["driver"] == "JP2OpenJPEG" and r.profile["transform"].c == pytest.approx(left)
    assert r.closed and r._data is None


def test_world_file_sidecar_and_identity_fallback(tmp_path):
    g = np.random.default_rng(5)
    arr = g.integers(0, 255, (30, 50, 3)).astype(np.uint8)
    p = str(tmp_path / "w.jp2")
    write_jp2(p, arr)
    with jp2.Jp2Raster(p) as r:  # no georeferencing at all: rasterio's identity (top = height, 1 unit per pixel)
        assert (r.left, r.top, r.res, r.crs) == (0.0, 30.0, (1.0, 1.0), None)
    # ESRI world file: the last two lines are the CENTRE of the top-left pixel
    open(str(tmp_path / "w.j2w"), "w").write("0.5\n0.0\n0.0\n-0.5\n1000.25\n1999.75\n")
    with jp2.Jp2Raster(p, default_crs="EPSG:2154") as r:
        assert (r.left, r.top, r.res, r.crs) == (1000.0, 2000.0, (0.5, 0.5), "EPSG:2154")
        assert np.array_equal(r.read(2), arr[:, :, 1])
    open(str(tmp_path / "w.j2w"), "w").write("0.5\n0.1\n0.0\n-0.5\n1000.25\n1999.75\n")
    with pytest.raises(jp2.Jp2Error, match="rotated"):
        jp2.Jp2Raster(p)


def test_embedded_box_wins_over_the_sidecar_and_raw_codestreams_open(tmp_path):
    from PIL import Image
    g = np.random.default_rng(6)
    arr = g.integers(0, 255, (20, 24)).astype(np.uint8)
    p = str(tmp_path / "e.jp2")
    write_jp2(p, arr, tmp_path, (10.0, 50.0, 2.0, "EPSG:4326"))
    open(str(tmp_path / "e.wld"), "w").write("1\n0\n0\n-1\n0.5\n19.5\n")
    with jp2.Jp2Raster(p) as r:
        assert (r.left, r.top, r.res, r.crs) == (10.0, 50.0, (2.0, 2.0), "EPSG:4326")
    k = str(tmp_path / "raw.j2k")
    Image.fromarray(arr).save(k, format="JPEG2000", irreversible=False, no_jp2=True)
    assert open(k, "rb").read(4) == jp2.J2K_SOC
    with open_raster(k) as r:
        assert np.array_equal(r.read(1), arr)


def test_bad_files_raise_with_the_reason(tmp_path):
    p = tmp_path / "x.jp2"
    p.write_bytes(jp2.JP2_SIGNATURE + struct.pack(">I4s", 4, b"ftyp") + b"\0" * 32)  # a box shorter than its header
    with pytest.raises(jp2.Jp2Error, match="malformed box"):
        jp2.Jp2Raster(str(p))
    q = tmp_path / "y.jp2"
    q.write_bytes(b"II*\0" + b"\0" * 32)
    assert not jp2.is_jpeg2000(str(q))
    with pytest.raises(jp2.Jp2Error, match="not a JPEG-2000"):
        jp2.Jp2Raster(str(q))
    with pytest.raises(GeoTiffError, match="not a TIFF"):  # the TIFF reader still says what it is not
        GeoTiffRaster(str(p))


def test_prep_config_picks_the_jp2_of_an_images_folder(tmp_path):
    """the fork's six-argument call: scripts/run_fast_aigle_segmentation.py:75-88 hands a folder of BD ORTHO tiles.  On
    this CPU-only box the call gets as far as the device check (the product has no CPU path); the full run from a .jp2
    is tests/test_zonal_gpu.py::test_zonal_run_on_a_jpeg2000_folder_equals_the_in_memory_run"""
    import torch
    import yaml
    from flair_zonal_detection.inference import prep_config
    from helpers import MOD, ROOT
    g = np.random.default_rng(7)
    folder = tmp_path / "ortho"
    folder.mkdir()
    write_jp2(str(folder / "tile_a.jp2"), g.integers(0, 255, (300, 410, 3)).astype(np.uint8), tmp_path,
              (651992.36, 6860417.84, 0.2, "EPSG:2154"))
    (folder / "notes.txt").write_text("not a raster")
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "golden", "zonal_config.yaml")))
    cfg["modalities"][MOD].update({"channels": [1, 2, 3],
                                   "normalization": {"type": "custom", "means": [100.0] * 3, "stds": [50.0] * 3}})
    cfg.update({"img_pixels_detection": 128, "margin": 16, "output_px_meters": 0.2})
    ck = tmp_path / "w.ckpt"
    ck.write_bytes(b"placeholder")  # validate_config only checks that it exists
    args = (cfg, str(ck), None, str(tmp_path / "out"), str(tmp_path / "log"))
    if torch.cuda.is_available():
        prep_config(*args, images_folder=str(folder))
    else:
        with pytest.raises(RuntimeError, match="MI355X"):
            prep_config(*args, images_folder=str(folder))
    assert cfg["modalities"][MOD]["input_img_path"].endswith("tile_a.jp2")
    assert cfg["image_bounds"] is not None and cfg["output_path"] == str(tmp_path / "out")
    with pytest.raises(FileNotFoundError, match="no raster"):
        prep_config(*args, images_folder=str(tmp_path / "log"))


def test_mosaics_beyond_pillows_pixel_guard_open_and_decode(tmp_path, monkeypatch):
    """A 25 000 x 25 000 BD ORTHO tile is 625 MP, 3.5x Pillow's MAX_IMAGE_PIXELS; Image.open itself raises
    DecompressionBombError above twice the limit.  Shrink the limit instead of growing the file: a 70 x 93 image is
    then "too large" in exactly the same way, in the constructor (header read) and in the decode."""
    from PIL import Image
    monkeypatch.setattr(Image, "MAX_IMAGE_PIXELS", 1000)
    g = np.random.default_rng(5)
    arr = g.integers(0, 255, (70, 93)).astype(np.uint8)
    p = str(tmp_path / "big.jp2")
    Image.fromarray(arr).save(p, format="JPEG2000", irreversible=False)
    with pytest.raises(Image.DecompressionBombError):
        Image.open(p)
    with open_raster(p) as r:
        assert (r.height, r.width) == arr.shape
        assert np.array_equal(r.read(1), arr)
    assert Image.MAX_IMAGE_PIXELS == 1000  # restored after both steps


@pytest.mark.parametrize("shape,tile", [((300, 421), (128, 128)), ((257, 130, 3), (64, 64)), ((200, 200, 4), (256, 64)),
                                        ((96, 80), (32, 32))])
def test_tiled_codestreams_are_decoded_tile_by_tile(tmp_path, shape, tile):
    """A tiled JPEG-2000 (BD ORTHO mosaics are) is read lazily: each window decodes only the code-stream tiles it
    touches, from one-tile streams cut out of the file (rewritten SIZ, tile index 0) -- pixel-identical to the whole-image
    decode, ragged last tiles and multi-tile windows included."""
    from PIL import Image
    g = np.random.default_rng(shape[0] + tile[0])
    arr = g.integers(0, 255, shape).astype(np.uint8)
    p = str(tmp_path / "t.jp2")
    Image.fromarray(arr).save(p, format="JPEG2000", irreversible=False, tile_size=tile)
    want = arr[None] if arr.ndim == 2 else arr.transpose(2, 0, 1)
    with open_raster(p) as r:
        ix = r.index
        assert r.lazy and (ix.xt, ix.yt) == tile and ix.ntx == -(-shape[1] // tile[0]) and ix.nty == -(-shape[0] // tile[1])
        assert sorted(ix.parts) == list(range(ix.ntx * ix.nty))
        # one interior window: only the tiles it touches get decoded
        wx, wy = min(tile[0], shape[1] - 12) - 5, min(tile[1], shape[0] - 12) - 3
        got = r.read(window=make_window(wx, wy, 11, 9))
        assert np.array_equal(got, want[:, wy:wy + 9, wx:wx + 11])
        touched = ((wx + 10) // tile[0] - wx // tile[0] + 1) * ((wy + 8) // tile[1] - wy // tile[1] + 1)
        assert len(r._tiles) == touched and r._data is None
        # boundless read over the bottom-right corner (ragged tiles) and the whole image
        assert np.array_equal(r.read(window=make_window(shape[1] - 20, shape[0] - 17, 40, 30), boundless=True),
                              ArrayRaster(want, 0.0, float(shape[0]), 1.0).read(window=make_window(shape[1] - 20, shape[0] - 17, 40, 30),
                                                                                boundless=True))
        assert np.array_equal(r.read(), want)
        r.prefetch_rows(0, tile[1])
        assert r._data is None  # the whole-image path was never taken
    assert not r._tiles


def test_streams_that_cannot_be_cut_fall_back_to_the_whole_decode(tmp_path):
    from PIL import Image
    g = np.random.default_rng(3)
    arr = g.integers(0, 255, (120, 150)).astype(np.uint8)
    for name, kw, why in (("one.jp2", {}, "single tile"), ("odd.jp2", {"tile_size": (48, 48)}, "power of two")):
        p = str(tmp_path / name)
        Image.fromarray(arr).save(p, format="JPEG2000", irreversible=False, **kw)
        with open_raster(p) as r:
            assert not r.lazy and why in r.index.why
            assert np.array_equal(r.read(1, window=make_window(40, 30, 70, 60)), arr[30:90, 40:110])
            assert r._data is not None
