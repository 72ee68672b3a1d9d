"""Worker for tests/test_distributed.py: one rank of a gloo (CPU) run of the row-tile shard +
gather + assembly logic of terminalraytracer_amd.distributed.  The per-rank renderer is the CPU
oracle here (checker standing in for the GPU so that the collective path is testable without
one); the product path, HipShardRenderer, plugs libtrt_hip.so into the same ShardedFrame."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import support as T  # noqa: E402
from terminalraytracer_amd import hip  # noqa: E402
from terminalraytracer_amd.distributed import ShardedFrame, shard_rows  # noqa: E402


def main():
    case_name, tile_rows = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    case = next(c for c in T.golden_cases(("small", "medium")) if c["name"] == case_name)
    scene = T.golden_scene(case)
    w, h, b, s = case["width"], case["height"], case["bounce_limit"], case["rays_per_pixel"]

    frame = ShardedFrame(w, h, rank, world, "cpu", tile_rows)
    rows = shard_rows(w, h, rank, world, tile_rows)
    assert frame.local_rows == len(rows) == hip.lib().trt_rowset_rows(C.byref(frame.rowset))
    shard = frame.shard.numpy()
    shard[:] = -1.0  # padding rows must never reach the frame
    # render the owned rows, one contiguous tile at a time
    i = 0
    while i < len(rows):
        j = i
        while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1:
            j += 1
        band, _ = T.oracle_render(scene, w, h, b, s, threads=2, rows=(rows[i], rows[j] + 1))
        shard[i:j + 1] = band
        i = j + 1
    for _ in range(2):  # the collective is re-entrant (one gather per frame)
        out = frame.assemble()
    if rank == 0:
        got = out.numpy()
        assert got.shape == (h, w, 3)
        assert T.fnv(got) == case["fb_fnv"], (T.fnv(got), case["fb_fnv"])
        print(f"DIST_OK world={world} tile_rows={tile_rows} {case_name} {T.fnv(got)}")
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
