"""Command-line entry of the zonal tile loop -- counterpart of the reference's flair_zonal_detection/main.py:8-14
(``python -m flair_zonal_detection.main --config <yaml>`` -> run_inference(config)).

Multi-GPU: the path shards by tiles with no collective.  Launched as one process per GPU,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m flair_zonal_detection.main --config zonal.yaml

every rank takes RANK / WORLD_SIZE / LOCAL_RANK from the environment, runs its contiguous slice of the tile grid on
its own GPU and writes ``<output>.r<rank>of<world>.tif`` (+ a one-band "written" mask); rank 0 then waits for the
part files of all ranks and joins them in rank order into the output the single-process run would have written
(``geotiff.merge_shard_files``: the reference's last-writer-wins for the clamped last row / column is preserved).
No process group is created: the only synchronisation is the appearance of the part files (each is renamed into
place when complete).
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
import time
from typing import Dict, Optional

sys.path.append(os.path.abspath(os.path.join(os.path.dirname(__file__), "../")))

logger = logging.getLogger(__name__)


def _remove_stale_parts(outputs) -> None:
    """called by run_inference once the outputs exist, before the tile loop: drop this rank's leftovers"""
    from flair_zonal_detection.geotiff import WRITTEN_SUFFIX
    for o in outputs.values():
        path = getattr(o, "path", None)
        if path:
            for f in (path, path + WRITTEN_SUFFIX):
                if os.path.exists(f):
                    os.remove(f)


def _wait_for(paths, timeout_s: float, newer_than: float = 0.0) -> None:
    t0 = time.time()
    while True:
        missing = [p for p in paths if not (os.path.exists(p) and os.path.getmtime(p) >= newer_than)]
        if not missing:
            return
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"part files of other ranks did not appear within {timeout_s:.0f} s: {missing[:4]}")
        time.sleep(0.2)


def run_sharded(config, rank: int, world: int, timeout_s: float = 3600.0, keep_parts: bool = False
                ) -> Optional[Dict[str, str]]:
    """This rank's share of a zonal run over ``world`` processes (one per GPU).  Rank 0 returns {task: merged path}
    once every rank's part file exists; the other ranks return None as soon as their own part is written."""
    import torch
    from flair_zonal_detection.geotiff import WRITTEN_SUFFIX, GeoTiffWriter, merge_shard_files
    from flair_zonal_detection.inference import run_inference
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count())
    # The only synchronisation between ranks is the appearance of part files, so files an earlier (crashed, or
    # --keep-parts) run left under the same names must not be mistaken for this run's: every rank removes its own
    # part + mask before it starts, and rank 0 only accepts parts at least as new as its own start.
    t_start = time.time()
    outputs = run_inference(config, shard=(rank, world), before_loop=_remove_stale_parts)
    for task, o in outputs.items():
        if not isinstance(o, GeoTiffWriter):
            raise TypeError("a sharded multi-process run needs file outputs (GeoTIFF paths), not in-memory rasters")
    if rank != 0:
        return None
    merged = {}
    for task, o in outputs.items():
        mine = o.path  # <base>.r0of<world>.tif
        base = mine[:-len(f".r0of{world}.tif")]
        parts = [f"{base}.r{r}of{world}.tif" for r in range(world)]
        # the mask is written after its part file; ranks start together (torchrun), so a mask older than this
        # rank's start (minus a generous clock / start-up skew) is a leftover that its owner has not replaced yet
        _wait_for([p + WRITTEN_SUFFIX for p in parts], timeout_s, newer_than=t_start - 600.0)
        merged[task] = merge_shard_files(parts, base + ".tif")
        if not keep_parts:
            for p in parts:
                for f in (p, p + WRITTEN_SUFFIX):
                    os.remove(f)
        logger.info("merged %d part files into %s", world, merged[task])
    return merged


def main(argv=None) -> None:
    parser = argparse.ArgumentParser(description="Run zonal detection inference.")
    parser.add_argument("--config", type=str, required=True, help="Path to the detection config file")
    parser.add_argument("--keep-parts", action="store_true", help="sharded runs: keep the per-rank part files")
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        run_sharded(args.config, int(os.environ.get("RANK", "0")), world, keep_parts=args.keep_parts)
    else:
        from flair_zonal_detection.inference import run_inference
        run_inference(args.config)


if __name__ == "__main__":
    main()
