"""CPU suite: the N > 1 path with world_size 2 on gloo.  Each rank produces the tile maps of its own frame (with the CPU
oracle standing in for the GPU encode, which needs a device), packs them in yk_export_tile_maps' layout, and the ranks run
exactly the collectives bench.py runs: the bbox all-reduce of the row-stripe layout and the ONE gather of tile maps."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.pyoracle import PASSES, OracleEncoder
from tests.images import edge_image, synth_planes
from yaik_amd import distributed as ykd


def _frame(rank):
    return synth_planes(256, n_planes=4, seed=12345 + rank) if rank == 0 else edge_image(256, 256, "mixed", 4, seed=rank)


def _encode_on_oracle(planes):
    enc = OracleEncoder(planes)
    m = enc.mip_prefilter()
    bitmaps = [enc.fitting_quad_smooth(sx, sy)[1] for sx, sy in PASSES]
    mask = enc.state("mipmapMask")                      # after the passes: 0 also where gradient tiles landed, so use the pre-pass rule
    n, h, w = planes.shape
    a = planes[3].reshape(h // 16, 16, w // 16, 16)
    keep = (a != 0).any(axis=(1, 3)).astype(np.uint8).ravel()
    defs, nibs, nn = [], [], []
    for p in range(3):
        d, nb, n_, _ = enc.dynamic_tile_encode(p, False)
        defs.append(d); nibs.append(nb); nn.append(n_)
    del mask
    return m, bitmaps, keep, defs, nibs, nn


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, bitmaps, keep, defs, nibs, nn = _encode_on_oracle(_frame(rank))
        payload, sizes = ykd.pack_blob(bitmaps, keep, defs, nibs, nn)
        # the row-stripe layout's only other exchange: image-wide kept-tile bbox
        stripe_box = np.array([16 * (rank + 1), 32 * (rank + 1), 200 - 8 * rank, 240 - 16 * rank], dtype=np.int32)
        gb = ykd.allreduce_bbox(stripe_box, dist, torch.device("cpu"))
        assert gb.tolist() == [16, 32, 200, 240]
        got = ykd.gather_tile_maps(payload, sizes, dist, torch.device("cpu"), dst=0)
        if rank == 0:
            assert got is not None and len(got) == world
            for r, (sz, pl) in enumerate(got):
                want = _encode_on_oracle(_frame(r))
                parts = ykd.split_blob(sz, pl)
                for i in range(7):
                    assert np.array_equal(parts["bitmaps"][i], want[1][i]), (r, i)
                assert np.array_equal(parts["keep"], want[2])
                for p in range(3):
                    assert np.array_equal(parts["defs"][p], want[3][p])
                    assert np.array_equal(parts["nibbles"][p], want[4][p])
                    assert parts["n_nibbles"][p] == want[5][p]
        else:
            assert got is None
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_world2_gather_of_tile_maps(oracle_built):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _pipe_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cap = 1 << 16
        pipe = ykd.TileMapGatherPipeline(dist, torch.device("cpu"), cap, dst=0)
        # payload sizes per step: steady, then a jump beyond the 12.5 % headroom on rank 1 (its transfer is repeated with the true length when
        # the step is retired), a bigger one on both ranks, then shrinking again
        sizes_per_step = [5000, 5100, 5050, 9000 if rank == 1 else 5000, 12000, 8000, 3000]
        results = []

        def payload(step, r, n):
            return ((np.arange(n, dtype=np.int64) * (r + 3) + step * 7) & 255).astype(np.uint8)

        def take(done):
            if done is not None or rank != 0:                  # payload views are only valid until the buffer is acquired again
                results.append(None if done is None else [(sz.copy(), pl.clone()) for sz, pl in done])
        for step, n in enumerate(sizes_per_step):
            buf, done = pipe.acquire()
            if step >= 2:
                take(done)
            sz = np.zeros(15, np.int64); sz[14] = n; sz[0] = step
            pipe.put(payload(step, rank, n), sz)               # header + sections, as yk_export_tile_maps_framed leaves them on the device
            pipe.submit()
        for done in pipe.flush():
            take(done)
        assert len(results) == len(sizes_per_step), len(results)
        # one grouped launch per step, plus one per repeated transfer: rank 1's payloads of step 3 (9000 against the 8192 agreed from step 1)
        # and of step 4 (12000 against the 8192 agreed from step 2; rank 0 is the root, its own payload never travels); step 5 fits the
        # count learnt from step 3
        if rank == 0:
            for step, res in enumerate(results):
                for r, (sz, pl) in enumerate(res):
                    n = 9000 if (step == 3 and r == 1) else sizes_per_step[step]
                    assert int(sz[14]) == n and int(sz[0]) == step, (step, r, sz)
                    assert np.array_equal(pl.numpy(), payload(step, r, n)), (step, r)
            assert pipe.regathers == 2, pipe.regathers
            assert pipe.collectives == len(sizes_per_step) + 2, pipe.collectives
        else:
            assert all(x is None for x in results)
            assert pipe.regathers == 2 and pipe.collectives == len(sizes_per_step) + 2, (pipe.regathers, pipe.collectives)
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_world2_pipelined_gather_with_regather():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_pack_split_roundtrip():
    rng = np.random.default_rng(5)
    bitmaps = [rng.integers(0, 256, n, dtype=np.uint8) for n in (2, 4, 4, 8, 16, 16, 32)]
    keep = rng.integers(0, 2, 37, dtype=np.uint8)
    defs = [rng.integers(0, 65536, n).astype(np.uint16) for n in (3, 0, 9)]
    nn = [7, 0, 18]
    nibs = [rng.integers(0, 256, (n + 1) // 2, dtype=np.uint8) for n in nn]
    payload, sizes = ykd.pack_blob(bitmaps, keep, defs, nibs, nn)
    assert payload.size % 16 == 0 and sizes[14] == payload.size
    parts = ykd.split_blob(sizes, payload)
    assert all(np.array_equal(a, b) for a, b in zip(parts["bitmaps"], bitmaps))
    assert np.array_equal(parts["keep"], keep)
    assert all(np.array_equal(a, b) for a, b in zip(parts["defs"], defs))
    assert all(np.array_equal(a, b) for a, b in zip(parts["nibbles"], nibs))
