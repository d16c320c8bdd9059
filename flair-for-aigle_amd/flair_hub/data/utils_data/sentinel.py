"""Host-side preparation of Sentinel time-series patches -- counterpart of the reference's
flair_hub/data/utils_data/sentinel.py (reshape_sentinel :7-17, filter_time_series :20-43, temporal_average :123-151 with
its monthly :46-77 and semi-monthly :80-120 branches), used by the zonal dataset (flair_zonal_detection/dataset.py:121-169).
Plain numpy; pinned by tests/golden/sentinel_utils.npz (outputs of the reference's own functions)."""
from __future__ import annotations

import datetime as _dt
from typing import Sequence, Tuple

import numpy as np


def reshape_sentinel(arr: np.ndarray, chunk_size: int = 10) -> np.ndarray:
    """[T * chunk, ...] band stack -> [T, chunk, ...] (bands of one date are consecutive in the raster)"""
    return arr.reshape((arr.shape[0] // chunk_size, chunk_size) + tuple(arr.shape[1:]))


def filter_time_series(data_array: np.ndarray, max_cloud_value: int = 1, max_snow_value: int = 1,
                       max_fraction_covered: float = 0.05) -> np.ndarray:
    """[T, 2 (snow, cloud), H, W] mask stack -> bool [T]: dates whose cloud- and snow-free share reaches
    1 - max_fraction_covered; when no date qualifies, the snow criterion alone decides"""
    clear = (data_array[:, 1] <= max_cloud_value) & (data_array[:, 0] <= max_snow_value)
    need = (1 - max_fraction_covered) * (data_array.shape[2] * data_array.shape[3])
    keep = clear.sum(axis=(1, 2)) >= need
    if not keep.any():
        keep = (data_array[:, 0] <= max_snow_value).sum(axis=(1, 2)) >= need
    return keep


def _as_datetimes(dates) -> list:
    out = []
    for d in list(dates):
        if isinstance(d, _dt.datetime):
            out.append(d)
        elif hasattr(d, "to_pydatetime"):
            out.append(d.to_pydatetime())
        else:
            out.append(_dt.datetime.utcfromtimestamp(np.datetime64(d, "s").astype("int64")))
    return out


def temporal_average(data: np.ndarray, dates: Sequence, period: str = "monthly", ref_date: str = "01-01"
                     ) -> Tuple[np.ndarray, np.ndarray]:
    """Period means of a [T, ...] series and the day offset of every period's middle from the reference day of the
    FIRST date's year: 12 months (membership by month number, whatever the year; middle = the 15th) or 24 half-months
    (1st-15th / 16th-end of the reference year; middles the 8th / 23rd).  An empty period repeats the previous one (zeros
    and offset 0 before the first non-empty period)."""
    if period not in ("monthly", "semi-monthly"):
        raise ValueError("Period must be either 'monthly' or 'semi-monthly'.")
    dts = _as_datetimes(dates)
    ref_month, ref_day = (int(v) for v in ref_date.split("-"))
    year = dts[0].year
    ref = _dt.datetime(year, ref_month, ref_day)
    groups = []  # (member indices, middle date)
    for month in range(1, 13):
        if period == "monthly":
            groups.append(([i for i, d in enumerate(dts) if d.month == month], _dt.datetime(year, month, 15)))
            continue
        nxt = _dt.datetime(year + 1, 1, 1) if month == 12 else _dt.datetime(year, month + 1, 1)
        for lo, hi, mid in ((_dt.datetime(year, month, 1), _dt.datetime(year, month, 15), 8),
                            (_dt.datetime(year, month, 16), nxt - _dt.timedelta(days=1), 23)):
            groups.append(([i for i, d in enumerate(dts) if lo <= d <= hi], _dt.datetime(year, month, mid)))
    means, offsets, last = [], [], None
    for idx, middle in groups:
        if idx:
            last = np.mean(data[idx], axis=0)
            means.append(last)
            offsets.append((middle - ref).days)
        else:
            means.append(last if last is not None else np.zeros_like(data[0]))
            offsets.append(offsets[-1] if offsets else 0)
    return np.array(means), np.array(offsets)
