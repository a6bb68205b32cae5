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
    """Image-wide bbox from per-rank stripe boxes: ONE tiny all-reduce (MAX over {-x0, -y0, x1, y1}).  Only the multi-process layout
    needs it; one process driving all GPUs combines the boxes on the host (EncoderContext::ConvertHotPathStripes)."""
    import torch
    v = torch.tensor([-int(bbox[0]), -int(bbox[1]), int(bbox[2]), int(bbox[3])], dtype=torch.int32, device=device)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    v = v.cpu().numpy()
    return np.array([-v[0], -v[1], v[2], v[3]], dtype=np.int32)


HEADER_BYTES = 128          # 16 x u64: payload bytes (the sections behind the header), then the 15 section sizes (yk_export_tile_maps_framed)


def frame_payload(payload: np.ndarray, sizes: np.ndarray) -> np.ndarray:
    """Host-side twin of yk_export_tile_maps_framed: header + sections."""
    hdr = np.zeros(16, dtype=np.uint64)
    hdr[0] = int(sizes[14])
    hdr[1:16] = np.asarray(sizes, dtype=np.uint64)
    return np.concatenate([hdr.view(np.uint8), np.asarray(payload, dtype=np.uint8)])


class TileMapGatherPipeline:
    """The ONE collective of the path: the gather of the per-rank tile maps onto rank `dst`, double-buffered for streams of frames / stripes.

    A payload is framed (a 128-byte header with its size table in front of the sections), so the root needs nothing but the bytes.  Payload
    sizes differ per rank and per frame, and both ends of a transfer must name the same count: the count for rank r at step k is derived,
    on rank r and on the root alike, from the header of rank r's payload of step k - 2 (+12.5 % headroom) -- the payload that used the same
    buffer, whose header both have read by then.  A step is therefore exactly one grouped launch of point-to-point transfers
    (torch.distributed.batch_isend_irecv = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on RCCL; every sender uses its own xGMI link
    into the root), with no size exchange next to it.  A payload that outgrows its agreed count arrives truncated; both ends see that in its
    header when the step is retired and repeat that one transfer with the true length.  One all_gather at set-up agrees the first counts.

    Per step:  buf, done = pipe.acquire()   # waits for the gather that last used this buffer (two steps ago), returns its result
               ... yk_export_tile_maps_framed into buf (or pipe.put(payload, sizes) for host data) ...
               pipe.submit()                # launches the transfers and returns immediately
    and `pipe.flush()` at the end.  The gather of step i (RCCL kernels on the communicator's stream) overlaps the encode kernels of step i+1.
    Results (on dst: list of (sizes[15], payload view) per rank; elsewhere None) stay valid until the buffer they came from is acquired again.
    """

    def __init__(self, dist, device, capacity: int, dst: int = 0, headroom: float = 1.125, staging_device=None):
        import torch
        self.dist, self.dst, self.headroom = dist, dst, headroom
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.comm_device = device
        self.cap = int(capacity) + HEADER_BYTES
        # staging_device: where the encoder writes (HBM); differs from the communicator's device only in the gloo rehearsal
        self.blobs = [torch.zeros(self.cap, dtype=torch.uint8, device=staging_device or device) for _ in range(2)]
        self.send = [None, None]
        self.recv = [[None] * self.world, [None] * self.world]
        self.count_used = [[0] * self.world, [0] * self.world]
        self.work = [None, None]
        self.agreed = None                                  # [world]: bytes rank r sends in its next transfer (known for all r on dst, for itself elsewhere)
        self.step = 0
        self.regathers = 0
        self.collectives = 0                                # grouped launches issued (set-up agreement excluded): one per step + one per repeated transfer

    def _roundup(self, n: int) -> int:
        return min(self.cap, (int(n * self.headroom) + 4095) & ~4095)

    def acquire(self):
        slot = self.step & 1
        return self.blobs[slot], self._retire(slot)

    def put(self, payload: np.ndarray, sizes: np.ndarray) -> None:
        """Host data into the acquired buffer (tests, CPU ranks): header + sections."""
        import torch
        framed = frame_payload(payload, sizes)
        buf = self.blobs[self.step & 1]
        buf[: framed.size] = torch.from_numpy(framed).to(buf.device)

    def _need(self, blob) -> int:
        import torch
        return HEADER_BYTES + int(blob[:8].cpu().view(torch.int64)[0].item())

    def submit(self) -> None:
        """Launch the gather of the acquired buffer (its header must be in place, written by the exporter on the device or by put())."""
        import torch
        slot = self.step & 1
        assert self.work[slot] is None, "acquire() the buffer before exporting into it"
        dist = self.dist
        if self.agreed is None:                             # set-up, once: every rank tells the length of its first payload
            mine = torch.tensor([self._need(self.blobs[slot])], dtype=torch.int64, device=self.comm_device)
            alln = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(alln, mine)
            self.agreed = [self._roundup(int(t.item())) for t in alln]
        ops, counts = [], list(self.agreed)
        if self.rank == self.dst:
            for r in range(self.world):
                if r == self.dst:
                    continue
                if self.recv[slot][r] is None or self.recv[slot][r].numel() < counts[r]:
                    self.recv[slot][r] = torch.empty(max(counts[r], HEADER_BYTES), dtype=torch.uint8, device=self.comm_device)
                ops.append(dist.P2POp(dist.irecv, self.recv[slot][r][: counts[r]], r))
        else:
            src = self.blobs[slot][: counts[self.rank]]
            if src.device != self.comm_device:
                src = src.to(self.comm_device)
            self.send[slot] = src                           # keep alive until retired
            ops.append(dist.P2POp(dist.isend, src, self.dst))
        self.work[slot] = dist.batch_isend_irecv(ops) if ops else []
        self.collectives += 1
        self.count_used[slot] = counts
        self.step += 1

    def _retire(self, slot: int):
        import torch
        if self.work[slot] is None:
            return None
        for w in self.work[slot]:
            w.wait()
        self.work[slot] = None
        dist, used = self.dist, self.count_used[slot]
        blob = self.blobs[slot]
        if self.rank != self.dst:
            need = self._need(blob)
            if need > used[self.rank]:                      # truncated: the root reads the same header and posts the matching receive
                self.regathers += 1
                src = blob[:need] if blob.device == self.comm_device else blob[:need].to(self.comm_device)
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, self.dst)]):
                    w.wait()
                self.collectives += 1
            self.agreed[self.rank] = self._roundup(need)
            return None
        out, again = [None] * self.world, []
        for r in range(self.world):
            buf = blob if r == self.dst else self.recv[slot][r]
            hdr = buf[:HEADER_BYTES].cpu().view(torch.int64).numpy()
            need = HEADER_BYTES + int(hdr[0])
            if r != self.dst and need > used[r]:
                if self.recv[slot][r].numel() < need:
                    self.recv[slot][r] = torch.empty(need, dtype=torch.uint8, device=self.comm_device)
                again.append(dist.P2POp(dist.irecv, self.recv[slot][r][:need], r))
            self.agreed[r] = self._roundup(need)
            out[r] = (hdr[1:16].copy(), r, need)
        if again:
            self.regathers += 1
            for w in dist.batch_isend_irecv(again):
                w.wait()
            self.collectives += 1
        res = []
        for sizes, r, need in out:
            buf = blob if r == self.dst else self.recv[slot][r]
            res.append((sizes, buf[HEADER_BYTES:need]))
        return res

    def flush(self) -> list:
        """Retire everything in flight, oldest first; returns their results in that order."""
        out = []
        for k in range(2):
            slot = (self.step + k) & 1
            if self.work[slot] is not None:
                out.append(self._retire(slot))
        return out


def gather_tile_maps(payload: np.ndarray, sizes: np.ndarray, dist, device, dst: int = 0):
    """One-shot form (host payload in yk_export_tile_maps' layout + its 15 sizes): set-up agreement + one grouped transfer.
    Returns on dst: list of (sizes[15], uint8 tensor of that rank's payload); else None."""
    cap = (int(sizes[14]) + 4095) & ~4095
    import torch
    c = torch.tensor([cap], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.MAX)
    pipe = TileMapGatherPipeline(dist, device, int(c.item()), dst=dst)
    pipe.acquire()
    pipe.put(payload, sizes)
    pipe.submit()
    return pipe.flush()[-1]


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
