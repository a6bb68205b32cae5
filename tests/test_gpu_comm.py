"""The C-ABI gather (yk_comm_* / yk_gather_maps, include/yaik_hip.h) on the devices this box has: a one-rank communicator always (the
framed export, RCCL bound by the library, the grouped launch on the handle's stream), all ranks of a one-process communicator when the box
has two or more GPUs."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.images import synth_planes
from yaik_amd import distributed as ykd
from yaik_amd._lib import lib
from yaik_amd.encoder import HipTileEncoder

pytestmark = pytest.mark.gpu


def _encode(dev, planes):
    enc = HipTileEncoder(dev)
    enc.set_image(planes)
    enc.mip_prefilter()
    enc.encode(3, False, False)
    return enc


def test_framed_export_header_matches_the_size_table():
    enc = _encode(0, synth_planes(512, n_planes=4))
    try:
        cap = enc.export_capacity()
        plain = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        sizes = enc.export_tile_maps(plain)
        framed = torch.zeros(cap + ykd.HEADER_BYTES, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        enc.export_tile_maps_framed(framed, None)
        enc.synchronize()
        hdr = framed[: ykd.HEADER_BYTES].cpu().view(torch.int64).numpy()
        assert int(hdr[0]) == int(sizes[14]) and np.array_equal(hdr[1:16].astype(np.uint64), sizes)
        assert torch.equal(framed[ykd.HEADER_BYTES: ykd.HEADER_BYTES + int(sizes[14])], plain[: int(sizes[14])])
    finally:
        enc.close()


def test_one_rank_communicator_gathers_its_own_payload():
    L = lib()
    assert L.yk_comm_available() == 1
    enc = _encode(0, synth_planes(256, n_planes=4))
    try:
        cap = enc.export_capacity() + ykd.HEADER_BYTES
        send = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        recv = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ident = (C.c_uint8 * 128)()
        assert L.yk_comm_unique_id(ident) == 0
        comm = C.c_void_p()
        assert L.yk_comm_init_rank(enc._h, ident, 1, 0, C.byref(comm)) == 0, L.yk_last_error(enc._h)
        n, me = C.c_int(), C.c_int()
        assert L.yk_comm_ranks(comm, C.byref(n), C.byref(me)) == 0 and (n.value, me.value) == (1, 0)
        enc.export_tile_maps_framed(send, None)             # no hand-over: the gather goes to the handle's own stream, behind the export
        rb, ro = (C.c_size_t * 1)(cap), (C.c_size_t * 1)(0)
        assert L.yk_gather_maps(enc._h, comm, 0, C.c_void_p(send.data_ptr()), cap, C.c_void_p(recv.data_ptr()), rb, ro) == 0, L.yk_last_error(enc._h)
        enc.synchronize()
        assert torch.equal(send, recv) and int(recv[:8].cpu().view(torch.int64)[0]) > 0
        L.yk_comm_destroy(comm)
    finally:
        enc.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_one_process_gather_over_all_devices():
    L = lib()
    n = min(torch.cuda.device_count(), 8)
    encs = [_encode(d, synth_planes(256, n_planes=4, seed=100 + d)) for d in range(n)]
    try:
        hs = (C.c_void_p * n)(*[e._h for e in encs])
        comms = (C.c_void_p * n)()
        assert L.yk_comm_init_all(hs, n, comms) == 0, L.yk_last_error(encs[0]._h)
        cap = encs[0].export_capacity() + ykd.HEADER_BYTES
        send = [torch.zeros(cap, dtype=torch.uint8, device=f"cuda:{d}") for d in range(n)]
        recv = torch.zeros(cap * n, dtype=torch.uint8, device="cuda:0")
        for d in range(n):
            torch.cuda.synchronize(d)
        for e, sbuf in zip(encs, send):
            e.export_tile_maps_framed(sbuf, None)
        sp = (C.c_void_p * n)(*[s.data_ptr() for s in send])
        sb = (C.c_size_t * n)(*[cap] * n)
        ro = (C.c_size_t * n)(*[cap * d for d in range(n)])
        assert L.yk_gather_maps_all(hs, comms, n, 0, sp, sb, C.c_void_p(recv.data_ptr()), ro) == 0, L.yk_last_error(encs[0]._h)
        for e in encs:
            e.synchronize()
        for d in range(n):
            assert torch.equal(recv[cap * d: cap * (d + 1)].cpu(), send[d].cpu()), d
        for d in range(n):
            L.yk_comm_destroy(comms[d])
    finally:
        for e in encs:
            e.close()
