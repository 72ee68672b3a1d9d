"""Row-tile sharding of one frame across the GPUs of a node, one process per GPU.

The frame producer is embarrassingly parallel over pixels; the only exchange is
assembling the framebuffer.  Rows are dealt in interleaved tiles of `tile_rows` rows (tile t ->
rank t mod world) so that sky rows and sphere/floor rows -- which differ ~10x in cost -- spread
evenly.  Every rank renders its tiles into a compact device buffer; ONE gather (RCCL over xGMI
when the backend is "nccl": each peer sends over its own direct link to the root) brings the
shards to rank 0, which scatters the rows into frame order with one index_copy.

`render_rows(camera, rowset, out_tensor)` is injectable so that the sharding/gather/assembly
logic is testable on CPU with the gloo backend; the product path is HipShardRenderer below.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import hip


def shard_rows(width, height, rank, world, tile_rows):
    """Frame rows owned by `rank`, ascending (same arithmetic as trt_rowset_frame_row)."""
    rs = hip.RowSet.shard(width, height, rank, world, tile_rows)
    n = hip.lib().trt_rowset_rows(C.byref(rs))
    return [hip.lib().trt_rowset_frame_row(C.byref(rs), i) for i in range(n)]


class ShardedFrame:
    """Owns the shard buffer of this rank, the gather buffers and the row permutation on the root."""

    def __init__(self, width, height, rank, world, device, tile_rows=8, root=0, dtype=torch.float64):
        """dtype float64: the reference's Screen layout (3 doubles per pixel, TerminalRayTracer.c:188-193);
        uint8: the (int)(c*255) bytes the emitter prints (TerminalRayTracer.c:1157-1163), 8x less to gather."""
        self.width, self.height, self.rank, self.world, self.root = width, height, rank, world, root
        self.device = torch.device(device)
        self.dtype = dtype
        self.rowset = hip.RowSet.shard(width, height, rank, world, tile_rows)
        per_rank = [shard_rows(width, height, r, world, tile_rows) for r in range(world)]
        self.local_rows = len(per_rank[rank])
        self.max_rows = max(len(r) for r in per_rank)  # shards are padded to equal size for the gather
        self.shard = torch.zeros((self.max_rows, width, 3), dtype=dtype, device=self.device)
        self.frame = None
        self.gathered = None
        if rank == root:
            # one contiguous buffer, the gather list is views into it: no concatenation per frame
            self.gathered_all = torch.zeros((world * self.max_rows, width, 3), dtype=dtype, device=self.device) if world > 1 else None
            self.gathered = list(self.gathered_all.split(self.max_rows, dim=0)) if world > 1 else None
            # row index in the concatenated (padded) gather buffer -> frame row
            src, dst = [], []
            for r, rows in enumerate(per_rank):
                src += [r * self.max_rows + i for i in range(len(rows))]
                dst += rows
            order = np.argsort(dst)
            assert sorted(dst) == list(range(height))
            self.take = torch.tensor(np.asarray(src)[order], dtype=torch.long, device=self.device)
            self.frame = torch.zeros((height, width, 3), dtype=dtype, device=self.device)

    def assemble(self):
        """Collective: gather every rank's shard to the root and put the rows in frame order.
        Returns the (H, W, 3) frame on the root, None elsewhere."""
        if self.world == 1:
            return self.shard[: self.height]  # a single renderer's rows are already in frame order
        if self.shard.is_cuda and dist.get_backend() == "gloo":
            # rehearsal path only (several ranks sharing one GPU, where RCCL cannot be used): stage through the host
            host = self.shard.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.root else None
            dist.gather(host, parts, dst=self.root)
            if self.rank != self.root:
                return None
            for dst, src in zip(self.gathered, parts):
                dst.copy_(src)
        else:
            dist.gather(self.shard, self.gathered if self.rank == self.root else None, dst=self.root)
            if self.rank != self.root:
                return None
        torch.index_select(self.gathered_all, 0, self.take, out=self.frame)
        return self.frame


class HipShardRenderer:
    """Product path: this rank's row tiles rendered by libtrt_hip.so into a shard tensor, frames PIPELINED.

    `depth` renderer contexts, each with its own HIP stream, shard buffer and gather buffers, take the frames
    round-robin.  For world > 1 use depth 3, and where a rank's frame is short (8 GPUs at 1080p) reserve some compute units
    (`reserve_cus`): the persistent workgroups of the next frame otherwise keep the gather's kernels (and this frame's
    reduction) off the machine until that frame has drained (tools/gather_sim.py: 0.37 -> 0.28 ms per 1/8 shard with a
    0.2 ms stand-in for the gather, 16 CUs reserved).  A persistent-wave frame ends with a tail in which most CUs are already idle; with two frames
    in flight the next frame's workgroups fill those CUs (measured on one MI355X at 1080p: 3.33 -> 2.98 ms per
    whole frame, 0.64 -> 0.38 ms per 1/8 shard), and the gather of frame f overlaps the rendering of f+1.
    Ordering is by events: a slot renders only after the assembly of its previous frame was enqueued, and the
    assembly waits for the slot's render.  The tensor `render` returns is valid on the current torch stream and
    is overwritten `depth` calls later."""

    def __init__(self, scene_data, width, height, rank, world, local_device, bounce_limit, rays_per_pixel,
                 tile_rows=8, depth=2, rgb8=False, reserve_cus=0):
        """rgb8=True: every rank quantises its shard on the device (trt_quantize_device) and the 3-byte pixels are
        gathered instead of the doubles -- all a terminal emitter needs, and bit-exact for it.
        reserve_cus > 0: the renderers' streams leave that many compute units alone (trt_reserve_cus).  The frame
        producer's workgroups are persistent and fill every wave slot; the gather's own kernels would otherwise have to
        wait for a frame to drain before they get on the machine."""
        torch.cuda.set_device(local_device)
        self.bounce_limit, self.rays_per_pixel = bounce_limit, rays_per_pixel
        self.rgb8 = rgb8
        self.slots = []
        for _ in range(max(1, depth)):
            ctx = hip.Context(local_device)
            ctx.set_scene(scene_data)
            if reserve_cus > 0:
                try:
                    ctx.reserve_cus(reserve_cus)  # re-creates the context's own stream with a CU mask
                except Exception as e:  # an optimisation only: render on the ordinary stream if the mask cannot be had
                    import sys
                    print(f"HipShardRenderer: no compute units reserved ({type(e).__name__}: {e})", file=sys.stderr)
                    reserve_cus = 0
            # the context's own stream, wrapped for torch's event calls: one stream per renderer and no more (every stream
            # of the process competes for a few hardware queues, and streams sharing a queue run one after the other)
            stream = torch.cuda.ExternalStream(ctx.stream_ptr(), device=torch.device(f"cuda:{local_device}"))
            frame = ShardedFrame(width, height, rank, world, f"cuda:{local_device}", tile_rows,
                                 dtype=torch.uint8 if rgb8 else torch.float64)
            self.slots.append({"ctx": ctx, "stream": stream, "frame": frame,
                               "pixels": torch.zeros((frame.max_rows, width, 3), dtype=torch.float64, device=f"cuda:{local_device}") if rgb8 else frame.shard,
                               "rendered": torch.cuda.Event(), "consumed": torch.cuda.Event()})
        self.external_streams = reserve_cus > 0  # CU-masked render streams
        # CU-masked streams are "blocking" streams: they order themselves against the NULL stream.  With them, the
        # assembly (wait, gather, index_select) runs on a stream of its own so that nothing in the loop touches stream 0.
        self.main = torch.cuda.Stream(device=local_device) if self.external_streams else None
        self.calls = 0
        self.ctx = self.slots[0]["ctx"]          # for counters / kernel selection helpers
        self.sharded = self.slots[0]["frame"]

    def for_each_context(self, fn):
        for slot in self.slots:
            fn(slot["ctx"])

    def render(self, camera):
        """Enqueue one frame; returns the assembled frame tensor on the root (None elsewhere).  The tensor is valid on the
        current torch stream -- or, when compute units are reserved, on `self.main` (synchronise the device, or make your
        stream wait on `self.main`, before reading it)."""
        slot = self.slots[self.calls % len(self.slots)]
        main = self.main if self.main is not None else torch.cuda.current_stream()
        if self.calls >= len(self.slots):
            slot["stream"].wait_event(slot["consumed"])  # the previous frame of this slot has been assembled
        self.calls += 1
        s = slot["frame"]
        px = slot["pixels"]
        slot["ctx"].render_device(camera, s.rowset, self.bounce_limit, self.rays_per_pixel, px.data_ptr(), px.numel() * 8)
        if self.rgb8:
            slot["ctx"].quantize_device(px.data_ptr(), s.local_rows * s.width, s.shard.data_ptr())
        slot["rendered"].record(slot["stream"])
        with torch.cuda.stream(main):
            main.wait_event(slot["rendered"])
            frame = s.assemble()
            slot["consumed"].record(main)
        return frame

    def kernel_times(self, launches):
        """HIP-event durations of the last `launches` render launches over all slots (they overlap in time)."""
        per = -(-launches // len(self.slots))
        out = []
        for slot in self.slots:
            out += slot["ctx"].kernel_times(per)
        return out[:launches] if len(out) > launches else out

    def close(self):
        torch.cuda.synchronize()
        for slot in self.slots:
            slot["ctx"].close()
