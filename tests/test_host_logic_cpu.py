"""Host-side bookkeeping of the product that needs no GPU (round-1 advisor findings)."""
import torch

from helpers import ROOT  # noqa: F401  (sets sys.path)


def test_batchnorm_pending_batch_count_does_not_survive_a_load():
    from flairhip.nn import HipBatchNorm2d
    bn = HipBatchNorm2d(8)
    bn.note_batch()
    bn.note_batch()  # two training batches counted lazily
    assert int(bn.state_dict()["num_batches_tracked"]) == 2
    bn.note_batch()
    sd = {k: v.clone() for k, v in bn.state_dict().items()}  # flushes: 3
    sd["num_batches_tracked"] = torch.tensor(40)
    bn.note_batch()  # pending again when the checkpoint arrives
    bn.load_state_dict(sd)
    assert int(bn.state_dict()["num_batches_tracked"]) == 40  # not 41: the loaded count is the truth
    bn.note_batch()
    assert int(bn.state_dict()["num_batches_tracked"]) == 41


def test_outgrown_workspaces_stay_alive_for_captured_graphs():
    from flairhip import ops
    dev = torch.device("cpu")
    a = ops.workspace(1 << 20, dev, "unit-test-slot")
    ptr = a.data_ptr()
    b = ops.workspace(4 << 20, dev, "unit-test-slot")
    assert b.numel() >= (4 << 20) and b.data_ptr() != ptr
    # the superseded buffer is still referenced (a replayed hipGraph may keep writing through its raw pointer)
    assert any(t.data_ptr() == ptr for t in ops._retired_workspaces)
    assert ops.workspace(1 << 20, dev, "unit-test-slot") is b  # grow-only


def test_nearest_bounds_read_and_time_series_tiles(tmp_path):
    """zonal dataset, time-series side (reference dataset.py:100-104,121-169): the nearest-neighbour mask read, the
    [T, C, h, w] reshape, per-tile cloud filtering without touching the shared date table, temporal averaging"""
    import numpy as np
    import pandas as pd
    from flair_zonal_detection.dataset import MultiModalSlicedDataset, pad_series_collate
    from flair_zonal_detection.raster import ArrayRaster
    rng = np.random.default_rng(1)
    T = 4
    data = rng.normal(size=(T * 10, 6, 8)).astype(np.float32)
    s2 = ArrayRaster(data, 1000.0, 2000.0, 10.0)
    # nearest: a 40 m box over 20 m pixels read at 4 x 4 -> every source pixel twice; outside the raster -> 0
    m = ArrayRaster(np.arange(2 * 3 * 4, dtype=np.uint8).reshape(2, 3, 4), 1000.0, 2000.0, 20.0)
    got = m.read_bounds([1, 2], (1000.0, 1960.0, 1040.0, 2000.0), 4, nearest=True)
    assert np.array_equal(got[0], np.repeat(np.repeat(m.data[0, :2, :2], 2, 0), 2, 1))
    edge = m.read_bounds([1], (1060.0, 1920.0, 1100.0, 1960.0), 2, nearest=True)  # bottom-right corner, half outside
    assert edge[0, 0, 0] == m.data[0, 2, 3] and edge[0, 1, 1] == 0 and edge[0, 0, 1] == 0
    masks = np.zeros((T * 2, 3, 4), np.uint8)
    masks[2 * 2 + 1] = 9            # date 2 cloudy everywhere
    masks[2 * 0 + 1, :, :2] = 9     # date 0 cloudy over the western half
    dates = tmp_path / "d.txt"
    dates.write_text("20210110\n20210125\n20210301\n20210620\n")
    cfg = {"SENTINEL2_TS": {"input_img_path": s2, "channels": list(range(1, 11)), "dates_txt": str(dates),
                            "filter_clouds": True, "filter_clouds_img_path": ArrayRaster(masks, 1000.0, 2000.0, 20.0)}}
    tiles = pd.DataFrame({"geometry": [(1000.0, 1960.0, 1040.0, 2000.0), (1040.0, 1940.0, 1080.0, 1980.0)]})
    ds = MultiModalSlicedDataset(tiles, cfg, {"SENTINEL2_TS": 4}, "05-15", {"labels": []})
    west, east = ds[0], ds[1]
    assert west["SENTINEL2_TS"].shape == (2, 10, 4, 4) and east["SENTINEL2_TS"].shape == (3, 10, 4, 4)
    # day offsets to 15 May of the same year; the west tile lost dates 0 and 2, the east tile date 2 only
    assert west["SENTINEL2_DATES"].tolist() == [-110.0, 36.0] and east["SENTINEL2_DATES"].tolist() == [-125.0, -110.0, 36.0]
    assert len(ds.series_dates["SENTINEL2_TS"]["dates"]) == 4  # the shared table is untouched (the reference shrinks it)
    # 1:1 read of the unfiltered bands: tile box = raster pixels [0:4, 0:4]
    assert np.array_equal(west["SENTINEL2_TS"][0].numpy(), data[10:20, :4, :4])
    batch = pad_series_collate([west, east])
    assert batch["SENTINEL2_TS"].shape == (2, 3, 10, 4, 4) and batch["SENTINEL2_DATES"].shape == (2, 3)
    assert float(batch["SENTINEL2_TS"][0, 2].abs().max()) == 0.0  # zero-padded date = what the U-TAE treats as padding
    cfg["SENTINEL2_TS"].update({"filter_clouds": False, "temporal_average": True})
    avg = MultiModalSlicedDataset(tiles, cfg, {"SENTINEL2_TS": 4}, "05-15", {"labels": []})[0]
    assert avg["SENTINEL2_TS"].shape == (12, 10, 4, 4) and avg["SENTINEL2_DATES"].shape == (12,)
    assert np.allclose(avg["SENTINEL2_TS"][0].numpy(), data[:20].reshape(2, 10, 6, 8)[:, :, :4, :4].mean(0), atol=1e-6)
