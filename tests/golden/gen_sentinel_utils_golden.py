#!/usr/bin/env python
"""tests/golden/sentinel_utils.npz: inputs and outputs of the reference's own Sentinel patch helpers
(/root/reference/flair_hub/data/utils_data/sentinel.py: numpy + pandas only, imports cleanly) -- reshape_sentinel,
filter_time_series (incl. the snow-only fallback), temporal_average monthly / semi-monthly (empty periods, dates spanning
two years).  Usage: python tests/golden/gen_sentinel_utils_golden.py"""
import datetime
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from flair_hub.data.utils_data.sentinel import filter_time_series, reshape_sentinel, temporal_average  # noqa: E402


def main():
    rng = np.random.default_rng(5)
    out = {}
    stack = rng.normal(size=(7 * 10, 6, 5)).astype(np.float32)
    out["reshape_in"], out["reshape_out"] = stack, reshape_sentinel(stack, chunk_size=10)
    masks = rng.integers(0, 100, size=(9, 2, 8, 8)).astype(np.uint8)
    masks[2] = 0
    masks[5, 1] = 0
    masks[5, 0, :1, :3] = 50  # 3 / 64 pixels of snow: still below the 5 % share
    out["filter_in"], out["filter_out"] = masks, filter_time_series(masks)
    cloudy = np.full((4, 2, 6, 6), 90, np.uint8)
    cloudy[1, 0] = 0  # clouds everywhere, one date without snow: the fallback keeps it
    out["filter2_in"], out["filter2_out"] = cloudy, filter_time_series(cloudy)
    days = ["20210105", "20210119", "20210203", "20210316", "20210317", "20210601", "20210630", "20211115", "20211201",
            "20220110"]
    dates = pd.Series([datetime.datetime.strptime(d, "%Y%m%d") for d in days])
    series = rng.normal(size=(len(days), 3, 4, 4)).astype(np.float32)
    out["avg_in"] = series
    out["avg_days"] = np.array([int(d) for d in days])
    for tag, period in (("m", "monthly"), ("s", "semi-monthly")):
        a, d = temporal_average(series, dates, period=period, ref_date="05-15")
        out[f"avg_{tag}_out"], out[f"avg_{tag}_days"] = a, d
    np.savez_compressed(os.path.join(HERE, "sentinel_utils.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
