"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The hot path shards with no data-path collective: units (16x16 macro-tiles, 8x8 tile-planes) are independent given
their pixels.  Two layouts are supported:
  * frame sharding  (batches: every rank encodes whole frames)            -> nothing but the final gather;
  * row stripes     (one large image: rank r owns rows [r*H/N, (r+1)*H/N), multiples of 64, + 1 halo row)
                    -> one tiny host min/max-combine of the alpha bounding boxes, then the final gather.
The only collective on the data path is ONE gather that concatenates the per-rank tile maps on the root.
Works with CPU tensors on gloo as well (used by the world_size-2 CPU tests).
"""
from __future__ import annotations

import numpy as np


def stripe_rows(full_h: int, world: int, rank: int) -> tuple[int, int, int]:
    """(y0, h, halo_rows) of rank's stripe.  The image's 64-row blocks are dealt as evenly as possible (floor + remainder), so every
    rank owns at least one block whenever there are at least `world` of them; stripe heights are multiples of 64 except the last
    one.  With fewer blocks than ranks the trailing ranks get h == 0: they skip the encode but must still join the bounding-box
    combine (with the empty box) and the gather (with an empty payload), see `stripe_is_empty`."""
    blocks = (full_h + 63) // 64
    base, rem = divmod(blocks, world)
    start = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    y0 = min(start * 64, full_h)
    y1 = min((start + count) * 64, full_h)
    halo = 1 if y1 < full_h and y1 > y0 else 0
    return y0, y1 - y0, halo


EMPTY_BBOX = (9999999, 9999999, -1, -1)


def stripe_is_empty(full_h: int, world: int, rank: int) -> bool:
    return stripe_rows(full_h, world, rank)[1] == 0


def combine_bboxes(boxes: np.ndarray) -> np.ndarray:
    """min/max-combine per-stripe kept-tile bounding boxes {x0,y0,x1,y1} (empty = {9999999,9999999,-1,-1})."""
    boxes = np.asarray(boxes, dtype=np.int64).reshape(-1, 4)
    return np.array([boxes[:, 0].min(), boxes[:, 1].min(), boxes[:, 2].max(), boxes[:, 3].max()], dtype=np.int32)


def allreduce_bbox(bbox: np.ndarray, dist, device) -> np.ndarray:
    """Image-wide bbox from per-rank stripe boxes: two tiny all-reduces (MIN on x0,y0 / MAX on x1,y1)."""
    import torch
    lo = torch.tensor([int(bbox[0]), int(bbox[1])], dtype=torch.int32, device=device)
    hi = torch.tensor([int(bbox[2]), int(bbox[3])], dtype=torch.int32, device=device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    lo = lo.cpu().numpy(); hi = hi.cpu().numpy()
    return np.array([lo[0], lo[1], hi[0], hi[1]], dtype=np.int32)


def gather_tile_maps(blob, nbytes: int, sizes: np.ndarray, dist, dst: int = 0):
    """ONE gather of the per-rank tile-map blobs (uint8 tensors, device or CPU) onto rank `dst`.

    Ranks first agree on the padded length (all_gather of 16 int64: payload bytes + the 15 section sizes), then
    gather equal-size slices.  Returns on dst: list of (sizes[15], uint8 tensor view of that rank's payload); else None.
    """
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    meta = torch.zeros(16, dtype=torch.int64, device=blob.device)
    meta[0] = int(nbytes)
    meta[1:16] = torch.as_tensor(np.asarray(sizes, dtype=np.int64), device=blob.device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = torch.stack(metas).cpu().numpy()
    pad = int(metas[:, 0].max())
    pad = (pad + 255) & ~255
    send = blob[:pad] if blob.numel() >= pad else torch.cat([blob, blob.new_zeros(pad - blob.numel())])
    if rank == dst:
        recv = [torch.empty(pad, dtype=torch.uint8, device=blob.device) for _ in range(world)]
        dist.gather(send, recv, dst=dst)
        return [(metas[r, 1:16].copy(), recv[r][: int(metas[r, 0])]) for r in range(world)]
    dist.gather(send, None, dst=dst)
    return None


class TileMapGatherPipeline:
    """Double-buffered, asynchronous form of gather_tile_maps for streams of frames / stripes.

    The gather of step i (RCCL kernels on the communicator's stream, xGMI links) overlaps the encode kernels of step i+1.
    Per step:  buf, done = pipe.acquire()   # waits for the gather that last used this buffer (two steps ago), returns its result
               ... export the tile maps into buf ...
               pipe.submit(nbytes, sizes)   # launches all_gather(sizes) + gather(payload) and returns immediately
    and `pipe.flush()` at the end.  All ranks use one padded length per gather, derived from the sizes every rank reported in
    an EARLIER step (+12.5 % headroom), so no rank waits for a size exchange before sending.  If some rank's payload outgrows
    that length the step is re-gathered with the safe synchronous protocol when it is retired (every rank sees the same size
    table, so all ranks take that branch together).  Results (on dst: list of (sizes[15], payload view) per rank; elsewhere
    None) stay valid until the buffer they came from is acquired again.
    """

    def __init__(self, dist, device, capacity: int, dst: int = 0, headroom: float = 1.125, staging_device=None):
        import torch
        self.dist, self.dst, self.headroom = dist, dst, headroom
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.comm_device = device
        # staging_device: where the encoder writes (HBM); differs from the communicator's device only in the gloo rehearsal
        self.blobs = [torch.empty(capacity, dtype=torch.uint8, device=staging_device or device) for _ in range(2)]
        self.send = [None, None]
        self.meta = [torch.zeros(16, dtype=torch.int64, device=device) for _ in range(2)]
        self.metas = [[torch.zeros(16, dtype=torch.int64, device=device) for _ in range(self.world)] for _ in range(2)]
        self.recv = [None, None]
        self.pad_used = [0, 0]
        self.work = [None, None]
        self.pad = 0
        self.step = 0
        self.regathers = 0

    def _roundup(self, n: int) -> int:
        return (int(n * self.headroom) + 4095) & ~4095

    def acquire(self):
        slot = self.step & 1
        return self.blobs[slot], self._retire(slot)

    def meta_tensor(self):
        """int64[16] device tensor of the current buffer: {payload bytes, sizes[0..14]} — the target of an asynchronous export."""
        return self.meta[self.step & 1]

    def submit(self, nbytes: int | None = None, sizes=None) -> None:
        """Launch the gather of the acquired buffer.  With (nbytes, sizes) the size table comes from the host; without, it was
        already written on the device (meta_tensor(), ordered before the communicator's stream by the exporter) — that form
        needs an agreed length, i.e. at least one earlier submit with host sizes."""
        import torch
        slot = self.step & 1
        assert self.work[slot] is None, "acquire() the buffer before exporting into it"
        dist = self.dist
        m = self.meta[slot]
        if nbytes is not None:
            m[0] = int(nbytes)
            m[1:16] = torch.as_tensor(np.asarray(sizes, dtype=np.int64), device=m.device)
        else:
            assert self.pad != 0, "the first submit of a pipeline needs the sizes on the host"
        if self.pad == 0:                                   # first step: agree on a length once, synchronously
            dist.all_gather(self.metas[slot], m)
            self.pad = self._roundup(int(torch.stack(self.metas[slot])[:, 0].max().item()))
            w_meta = None
        else:
            w_meta = dist.all_gather(self.metas[slot], m, async_op=True)
        pad = min(self.pad, self.blobs[slot].numel())
        src = self.blobs[slot][:pad]
        if src.device != self.comm_device:
            src = src.to(self.comm_device)
        self.send[slot] = src                               # keep alive until retired
        if self.rank == self.dst:
            if self.recv[slot] is None or self.recv[slot][0].numel() != pad:
                self.recv[slot] = [torch.empty(pad, dtype=torch.uint8, device=self.comm_device) for _ in range(self.world)]
            w = dist.gather(src, self.recv[slot], dst=self.dst, async_op=True)
        else:
            w = dist.gather(src, None, dst=self.dst, async_op=True)
        self.work[slot] = (w_meta, w)
        self.pad_used[slot] = pad
        self.step += 1

    def _retire(self, slot: int):
        import torch
        if self.work[slot] is None:
            return None
        w_meta, w = self.work[slot]
        if w_meta is not None:
            w_meta.wait()
        w.wait()
        self.work[slot] = None
        metas = torch.stack(self.metas[slot]).cpu().numpy()
        need = int(metas[:, 0].max())
        pad = self.pad_used[slot]
        self.pad = max(self.pad, self._roundup(need))
        if need > pad:                                      # a payload was truncated: repeat this step with the safe protocol
            self.regathers += 1
            blob = self.blobs[slot]
            if blob.device != self.comm_device:
                blob = blob.to(self.comm_device)
            return gather_tile_maps(blob, int(metas[self.rank, 0]), metas[self.rank, 1:16], self.dist, dst=self.dst)
        if self.rank != self.dst:
            return None
        return [(metas[r, 1:16].copy(), self.recv[slot][r][: int(metas[r, 0])]) for r in range(self.world)]

    def flush(self) -> list:
        """Retire everything in flight, oldest first; returns their results in that order."""
        out = []
        for k in range(2):
            slot = (self.step + k) & 1
            if self.work[slot] is not None:
                out.append(self._retire(slot))
        return out


def pack_blob(bitmaps, keep, defs, nibbles, n_nibbles) -> tuple[np.ndarray, np.ndarray]:
    """Host-side twin of yk_export_tile_maps' layout (sections padded to 16 bytes); returns (payload uint8, sizes[15])."""
    sizes = np.zeros(15, dtype=np.uint64)
    chunks = []

    def put(a):
        b = np.ascontiguousarray(a).view(np.uint8).ravel()
        chunks.append(b)
        pad = (-b.size) & 15
        if pad:
            chunks.append(np.zeros(pad, np.uint8))
    for i in range(7):
        sizes[i] = bitmaps[i].size
        put(bitmaps[i])
    sizes[7] = keep.size
    put(keep)
    for p in range(3):
        sizes[8 + 2 * p] = defs[p].size
        sizes[9 + 2 * p] = n_nibbles[p]
        put(np.asarray(defs[p], dtype=np.uint16))
        put(np.asarray(nibbles[p], dtype=np.uint8)[: (n_nibbles[p] + 1) // 2])
    payload = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    sizes[14] = payload.size
    return payload, sizes


def split_blob(sizes: np.ndarray, payload) -> dict:
    """Inverse of yk_export_tile_maps' layout: dict of numpy arrays from one rank's payload (CPU uint8 array/tensor)."""
    buf = payload.cpu().numpy() if hasattr(payload, "cpu") else np.asarray(payload, dtype=np.uint8)
    out, off = {}, 0

    def take(n):
        nonlocal off
        v = buf[off: off + n]
        off += (n + 15) & ~15
        return v
    out["bitmaps"] = [take(int(sizes[i])).copy() for i in range(7)]
    out["keep"] = take(int(sizes[7])).copy()
    out["defs"], out["nibbles"], out["n_nibbles"] = [], [], []
    for p in range(3):
        nd, nn = int(sizes[8 + 2 * p]), int(sizes[9 + 2 * p])
        out["defs"].append(take(nd * 2).copy().view(np.uint16))
        out["nibbles"].append(take((nn + 1) // 2).copy())
        out["n_nibbles"].append(nn)
    return out


def merge_corner_streams(streams: list[list[np.ndarray]], edges: list[tuple[np.ndarray, np.ndarray]]) -> list[np.ndarray]:
    """Image-wide corner-colour streams (`rgbStream` of the 7 gradient passes) from per-stripe streams.

    streams[s][p] = stripe s's stream of pass p (3 bytes per corner, de-duplicated inside the stripe);
    edges[s] = (keys[2, n], index[2, n]) from yk_gradient_corner_edges: first / last lattice row of stripe s.
    A lattice point on the boundary between stripes s and s+1 may have been emitted by both.  The reference emits it once,
    at its first toucher in (pass, scan order); tiles of stripe s precede those of stripe s+1 inside a pass, so the copy
    of the later pass is dropped, and stripe s+1's copy on a tie.  Returns the 7 concatenated streams.
    """
    n_stripes = len(streams)
    drop = [[[] for _ in range(7)] for _ in range(n_stripes)]
    NONE = np.uint32(0xFFFFFFFF)
    for s in range(n_stripes - 1):
        ka, ia = edges[s][0][1], edges[s][1][1]             # last row of stripe s
        kb, ib = edges[s + 1][0][0], edges[s + 1][1][0]     # first row of stripe s+1
        both = (ka != NONE) & (kb != NONE)
        pa, pb = (ka >> 27).astype(np.int64), (kb >> 27).astype(np.int64)
        for x in np.nonzero(both)[0]:
            if pa[x] <= pb[x]:
                drop[s + 1][int(pb[x])].append(int(ib[x]))
            else:
                drop[s][int(pa[x])].append(int(ia[x]))
    out = []
    for p in range(7):
        parts = []
        for s in range(n_stripes):
            st = np.asarray(streams[s][p], dtype=np.uint8).reshape(-1, 3)
            if drop[s][p]:
                keep = np.ones(st.shape[0], dtype=bool)
                keep[np.asarray(drop[s][p], dtype=np.int64)] = False
                st = st[keep]
            parts.append(st.reshape(-1))
        out.append(np.concatenate(parts) if parts else np.zeros(0, np.uint8))
    return out


def concat_nibble_streams(streams: list[np.ndarray], counts: list[int]) -> tuple[np.ndarray, int]:
    """Concatenate per-stripe nibble streams (low nibble first) into the image-wide stream of DynamicTileEncode."""
    total = int(sum(counts))
    out = np.zeros((total + 1) // 2, dtype=np.uint8)
    pos = 0
    for s, n in zip(streams, counts):
        if n == 0:
            continue
        nib = np.empty(2 * len(s), dtype=np.uint8)
        nib[0::2] = s & 15
        nib[1::2] = s >> 4
        nib = nib[:n]
        idx = pos + np.arange(n)
        np.bitwise_or.at(out, idx >> 1, (nib << ((idx & 1) * 4)).astype(np.uint8))
        pos += n
    return out, total
