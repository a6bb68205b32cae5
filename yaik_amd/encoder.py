"""Python plumbing over the C-ABI (include/yaik_hip.h) for tests and bench.py.

The method names follow the reference's operator surface for this path
(``EncoderContext::MipPrefilter / FittingQuadSmooth / DynamicTileEncode``, encoder/EncoderContext.h:326-370):
one fused launch computes what the seven FittingQuadSmooth calls and the three DynamicTileEncode calls
compute; the per-call views below hand out the cached results.  The C++ mirror of the same surface lives in
yaik_amd/host/ (the reference is compiled C++, so that is the drop-in; this module only moves bytes).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import YaikError, lib

PASSES = [(4, 4), (4, 3), (3, 4), (3, 3), (3, 2), (2, 3), (2, 2)]   # EncoderContext.cpp:9057-9093


# handle -> the library build that created it (the product library, or the test-hooks build of the same sources: tests/csrc/libyaik_hip_test.so):
# a handle is only ever passed back to the build that owns it, error lookups included
_HANDLE_LIB: dict = {}


def _chk(h, rc: int, L=None):
    if rc != 0:
        owner = L or _HANDLE_LIB.get(getattr(h, "value", h)) or lib()
        msg = owner.yk_last_error(h)
        raise YaikError(f"yaik_hip error {rc}: {msg.decode() if msg else '?'}")


class HipTileEncoder:
    """One handle = one GPU = one image or one row stripe of an image."""

    def __init__(self, device: int = 0, hooks: bool = False):
        """hooks=True: a handle of the TEST build of the library (include/yaik_hip_test.h: self-tests, ablations, cross-check kernel) --
        test infrastructure; the product path never asks for it."""
        from ._lib import test_lib
        L = test_lib() if hooks else lib()
        self._L = L
        h = C.c_void_p()
        rc = L.yk_create(device, C.byref(h))
        if rc != 0:
            raise YaikError(f"yk_create failed ({rc}): no usable HIP device -- the product path has no CPU fallback")
        self._h = h
        _HANDLE_LIB[h.value] = L
        self._keepalive = None
        self.w = self.h = self.n = 0

    def close(self):
        if getattr(self, "_h", None):
            _HANDLE_LIB.pop(self._h.value, None)
            self._L.yk_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    # ---- EncoderContext::SetImageToEncode ----------------------------------------------------------
    def set_image(self, planes, full_h: int | None = None, y0: int = 0, halo_rows: int = 0):
        """planes: numpy int32 [n, rows, w] (uploaded) or torch int32 cuda tensor [n, rows, w] (bound in place).
        rows = owned rows + halo_rows."""
        L = self._L
        is_torch = hasattr(planes, "data_ptr")
        n, rows, w = planes.shape
        h = rows - halo_rows
        self.n, self.h, self.w = n, h, w
        self.full_h = full_h if full_h is not None else h
        self.y0 = y0
        _chk(self._h, L.yk_set_image(self._h, w, self.full_h, n, y0, h, halo_rows))
        if is_torch:
            import torch
            assert planes.dtype == torch.int32 and planes.is_cuda and planes.is_contiguous()
            base = planes.data_ptr()
            ptrs = (C.c_void_p * 4)(*[base + i * rows * w * 4 if i < n else None for i in range(4)])
            # The handle launches on its own non-blocking stream, which is not ordered against torch's streams (measured: a kernel
            # launched here right after a torch kernel reads stale planes).  The hand-over is therefore a host-side fence; results
            # handed back to torch are fenced the same way (every getter and yk_export_tile_maps synchronise the handle's stream).
            # (_lib.lib() brings torch's bundled HIP runtime up first, so that this library binds to the same runtime instance.)
            torch.cuda.current_stream(planes.device).synchronize()
            _chk(self._h, L.yk_bind_device_planes(self._h, ptrs, w))
        else:
            planes = np.ascontiguousarray(planes, dtype=np.int32)
            ptrs = (C.c_void_p * 4)(*[planes[i].ctypes.data if i < n else None for i in range(4)])
            _chk(self._h, L.yk_upload_planes(self._h, ptrs, w))
            _chk(self._h, L.yk_synchronize(self._h))
        self._keepalive = planes

    def validate_planes(self) -> int:
        """Samples of the bound planes outside 0..255 (the precondition of the path; yk_upload_planes enforces it itself)."""
        n = C.c_size_t(0)
        _chk(self._h, self._L.yk_validate_planes(self._h, C.byref(n)))
        return int(n.value)

    # ---- EncoderContext::MipPrefilter ---------------------------------------------------------------
    def alpha_reject(self):
        _chk(self._h, self._L.yk_alpha_reject(self._h))

    def stripe_bbox(self) -> np.ndarray:
        b = np.zeros(4, dtype=np.int32)
        _chk(self._h, self._L.yk_get_stripe_bbox(self._h, b.ctypes.data))
        return b

    def alpha_finish(self, global_bbox: np.ndarray | None = None):
        if global_bbox is None:
            _chk(self._h, self._L.yk_alpha_finish(self._h, None))
        else:
            g = np.ascontiguousarray(global_bbox, dtype=np.int32)
            _chk(self._h, self._L.yk_alpha_finish(self._h, g.ctypes.data))

    def mip_prefilter(self) -> dict:
        """Whole-image EncoderContext::MipPrefilter: reject + finish + results."""
        if self.n == 4:
            self.alpha_reject()
            self.alpha_finish(None)
        return self.alpha_result()

    def alpha_result(self) -> dict:
        L = self._L
        b = np.zeros(4, dtype=np.int32); tb = np.zeros(4, dtype=np.int32)
        has, rem = C.c_int(), C.c_int()
        _chk(self._h, L.yk_alpha_result(self._h, b.ctypes.data, C.byref(has), C.byref(rem), tb.ctypes.data))
        out = np.zeros(max(1, (self.w // 16) * (self.full_h // 16) // 8 + 8), dtype=np.uint8)
        nb = C.c_size_t()
        _chk(self._h, L.yk_alpha_bitmap(self._h, out.ctypes.data, out.size, C.byref(nb)))
        return {"has_chunk": bool(has.value), "bounds": b, "remaining": rem.value, "tile_bbox": tb, "bitmap": out[:nb.value].copy()}

    # ---- 7x FittingQuadSmooth + 3x DynamicTileEncode, one launch ---------------------------------------
    def encode(self, reject_factor: int = 3, mode3bit_only: bool = False, want_dst: bool = False, dst_fill: int = -1):
        L = self._L
        _chk(self._h, L.yk_set_dst_fill(self._h, dst_fill))
        _chk(self._h, L.yk_encode_tiles(self._h, reject_factor, int(mode3bit_only), int(want_dst)))

    def set_batch(self, frames):
        """frames: torch int32 cuda tensor [F, n, h, w] (contiguous): F equally shaped images bound in place (yk_set_batch)."""
        import torch
        L = self._L
        assert frames.dtype == torch.int32 and frames.is_cuda and frames.is_contiguous() and frames.dim() == 4
        F, n, h, w = frames.shape
        self.n, self.h, self.w, self.full_h, self.y0 = n, h, w, h, 0
        _chk(self._h, L.yk_set_image(self._h, w, h, n, 0, h, 0))
        _chk(self._h, L.yk_set_batch(self._h, F))
        torch.cuda.current_stream(frames.device).synchronize()          # hand-over fence, see set_image
        base = frames.data_ptr()
        ptrs = (C.c_void_p * 4)(*[base + i * h * w * 4 if i < n else None for i in range(4)])
        _chk(self._h, L.yk_bind_device_batch(self._h, ptrs, w, n * h * w))
        self._keepalive = frames

    def encode_batch(self, reject_factor: int = 3, mode3bit_only: bool = False):
        _chk(self._h, self._L.yk_encode_batch(self._h, reject_factor, int(mode3bit_only)))

    def order_fused_after(self, other: "HipTileEncoder"):
        """The next encode() of this handle starts its fused kernel after the fused kernel last launched on `other` has finished
        (device-side wait; see yk_order_fused_after)."""
        _chk(self._h, self._L.yk_order_fused_after(self._h, other._h))

    def set_pixel_cache(self, on: bool = True):
        """From the next encode() on, the fused kernel leaves the packed pixels of the cells it did not cover for the live 1-D path
        (dynamic_tile_compressor), which then reads 4 B per pixel from there instead of 12 B from the planes (yk_set_pixel_cache)."""
        _chk(self._h, self._L.yk_set_pixel_cache(self._h, 1 if on else 0))

    def select_frame(self, f: int):
        _chk(self._h, self._L.yk_select_frame(self._h, f))

    def encode_frame(self, reject_factor: int = 3, mode3bit_only: bool = False):
        """alpha reject + fused kernel + compaction as one replayed hipGraph launch (whole images; see yk_encode_frame)."""
        _chk(self._h, self._L.yk_encode_frame(self._h, reject_factor, int(mode3bit_only)))

    def synchronize(self):
        _chk(self._h, self._L.yk_synchronize(self._h))

    def gradient_bitmap(self, p: int) -> np.ndarray:
        L = self._L
        n = L.yk_gradient_bitmap_bytes(self._h, p)
        out = np.zeros(n, dtype=np.uint8)
        _chk(self._h, L.yk_gradient_bitmap(self._h, p, out.ctypes.data, n))
        return out

    def gradient_counts(self) -> np.ndarray:
        c = np.zeros(7, dtype=np.int32)
        _chk(self._h, self._L.yk_gradient_counts(self._h, c.ctypes.data))
        return c

    def coverage(self) -> np.ndarray:
        """[h/4, w/4] bool: 4x4 cell covered by an accepted gradient tile (smoothMap != 0)."""
        mtw, mth = (self.w + 15) // 16, (self.h + 15) // 16
        raw = np.zeros(mtw * mth, dtype=np.uint16)
        _chk(self._h, self._L.yk_coverage(self._h, raw.ctypes.data, raw.size))
        bits = (raw.reshape(mth, mtw, 1) >> np.arange(16, dtype=np.uint16)) & 1
        cells = bits.reshape(mth, mtw, 4, 4).transpose(0, 2, 1, 3).reshape(mth * 4, mtw * 4)
        return cells[: self.h // 4, : self.w // 4].astype(bool)

    def gradient_corners(self, p: int) -> np.ndarray:
        cap = (self.w // 4 + 1) * (self.h // 4 + 2) * 3 + 16
        out = np.zeros(cap, dtype=np.uint8)
        nb = C.c_size_t()
        _chk(self._h, self._L.yk_gradient_corners(self._h, p, out.ctypes.data, cap, C.byref(nb)))
        return out[:nb.value].copy()

    # ---- FittingQuadSmooth with NULL planes (EncoderContext.cpp:3710, call sites :9261-9415) ---------------
    def fitting_quad_smooth_planes(self, plane_bit: int, sx: int = 2, sy: int = 2, reject_factor: int = 3):
        """One more gradient pass over the planes of `plane_bit` (bit0/1/2 = R/G/B present), after encode_tiles().
        Returns (tiles accepted, swizzled bitmap bytes, corner stream bytes) like the reference's TileDone / pFillBitMap / rgbStream."""
        L = self._L
        n = C.c_int()
        _chk(self._h, L.yk_gradient_partial_pass(self._h, reject_factor, plane_bit, sx, sy, C.byref(n)))
        nb = C.c_size_t()
        _chk(self._h, L.yk_partial_bitmap(self._h, None, 0, C.byref(nb)))
        bm = np.zeros(nb.value, dtype=np.uint8)
        if bm.size:
            _chk(self._h, L.yk_partial_bitmap(self._h, bm.ctypes.data, bm.size, None))
        _chk(self._h, L.yk_partial_corners(self._h, None, 0, C.byref(nb)))
        cs = np.zeros(nb.value, dtype=np.uint8)
        if cs.size:
            _chk(self._h, L.yk_partial_corners(self._h, cs.ctypes.data, cs.size, None))
        return int(n.value), bm, cs

    def gradient_preview(self, passes=range(7)) -> np.ndarray:
        """FittingQuadSmooth's testOutput planes [3, h, w] int32 after the given passes (7 = the last plane-subset pass); INT32_MIN = untouched."""
        out = np.zeros((3, self.h, self.w), dtype=np.int32)
        for p in passes:
            _chk(self._h, self._L.yk_gradient_preview(self._h, int(p), out.ctypes.data, out.size))
        return out

    def coverage_plane(self, plane: int) -> np.ndarray:
        """[h/4, w/4] bool: 4x4 cell of `plane` covered by an accepted tile (mapSmoothTile[plane] != 0)."""
        mtw, mth = (self.w + 15) // 16, (self.h + 15) // 16
        raw = np.zeros(mtw * mth, dtype=np.uint16)
        _chk(self._h, self._L.yk_coverage_plane(self._h, plane, raw.ctypes.data, raw.size))
        bits = (raw.reshape(mth, mtw, 1) >> np.arange(16, dtype=np.uint16)) & 1
        cells = bits.reshape(mth, mtw, 4, 4).transpose(0, 2, 1, 3).reshape(mth * 4, mtw * 4)
        return cells[: self.h // 4, : self.w // 4].astype(bool)

    # ---- (f)4 3-D LUT tiles (Load3DPattern / StartCorrelationSearch / Correlation3DSearch) ----------------------------
    def lut_clear(self) -> None:
        _chk(self._h, self._L.yk_lut_clear(self._h))

    def lut_load(self, pattern: np.ndarray) -> int:
        """pattern: uint8 [count, 3] with 6-bit coordinates (one Bank3D .lut file).  Returns the pattern's number."""
        p = np.ascontiguousarray(pattern, dtype=np.uint8)
        r, g, b = (np.ascontiguousarray(p[:, k]) for k in range(3))
        idx = C.c_int()
        _chk(self._h, self._L.yk_lut_load_pattern(self._h, r.ctypes.data, g.ctypes.data, b.ctypes.data, len(p), C.byref(idx)))
        return int(idx.value)

    def lut_tables(self, k: int):
        fac = np.zeros((4, 3, 64), np.int16); dist = np.zeros(64 ** 3, np.uint16); pos = np.zeros((4, 64 ** 3), np.uint8)
        _chk(self._h, self._L.yk_lut_pattern_tables(self._h, k, fac.ctypes.data, dist.ctypes.data, pos.ctypes.data))
        return fac, dist, pos

    def lut_start(self) -> None:
        _chk(self._h, self._L.yk_lut_start(self._h))

    def lut_search(self, sx: int, sy: int) -> int:
        n = C.c_int()
        _chk(self._h, self._L.yk_lut_search(self._h, sx, sy, C.byref(n)))
        return int(n.value)

    def lut_streams(self) -> dict:
        L = self._L
        out = {}
        names = ["tileType", "color", "idx3", "idx4", "idx5", "idx6"] + [f"map{k}" for k in range(6)]
        for which, name in enumerate(names):
            nb = C.c_size_t()
            _chk(self._h, L.yk_lut_stream(self._h, which, None, 0, C.byref(nb)))
            buf = np.zeros(nb.value, np.uint8)
            if buf.size:
                _chk(self._h, L.yk_lut_stream(self._h, which, buf.ctypes.data, buf.size, None))
            out[name] = buf.view(np.uint16) if name == "tileType" else buf
        return out

    def gradient_corner_edges(self):
        """(keys[2, w/4+1], index[2, w/4+1]) of the stripe's first and last lattice rows (see yk_gradient_corner_edges)."""
        n = self.w // 4 + 1
        keys = np.zeros((2, n), dtype=np.uint32); idx = np.zeros((2, n), dtype=np.uint32)
        _chk(self._h, self._L.yk_gradient_corner_edges(self._h, keys.ctypes.data, idx.ctypes.data, 2 * n))
        return keys, idx

    def range_streams(self, plane: int):
        L = self._L
        nd, nn = C.c_size_t(), C.c_size_t()
        _chk(self._h, L.yk_range_sizes(self._h, plane, C.byref(nd), C.byref(nn)))
        defs = np.zeros(nd.value, dtype=np.uint16)
        nib = np.zeros((nn.value + 1) // 2, dtype=np.uint8)
        _chk(self._h, L.yk_range_streams(self._h, plane, defs.ctypes.data if defs.size else None, defs.size,
                                          nib.ctypes.data if nib.size else None, nib.size))
        return defs, nib, nn.value

    def range_dst(self, plane: int) -> np.ndarray:
        out = np.zeros((self.h, self.w), dtype=np.int32)
        _chk(self._h, self._L.yk_range_dst(self._h, plane, out.ctypes.data, out.size))
        return out

    # ---- 3x DynamicTileCompressor (live 1-D range path, '1DTL') -----------------------------------------
    def dynamic_tile_compressor(self):
        """Returns (pix_stream, type_stream) exactly as GenerateDynamicTileChunk receives them."""
        L = self._L
        _chk(self._h, L.yk_range1d_encode(self._h))
        npx, nty = C.c_size_t(), C.c_size_t()
        _chk(self._h, L.yk_range1d_streams(self._h, None, 0, C.byref(npx), None, 0, C.byref(nty)))
        pix = np.zeros(npx.value, dtype=np.uint8); typ = np.zeros(nty.value, dtype=np.uint8)
        _chk(self._h, L.yk_range1d_streams(self._h, pix.ctypes.data if pix.size else None, pix.size, None,
                                            typ.ctypes.data if typ.size else None, typ.size, None))
        return pix, typ

    def gradient_corners_run(self) -> None:
        """Builds the seven corner streams on the device (no copy to the host)."""
        _chk(self._h, self._L.yk_gradient_corners_run(self._h))

    def stage_ms(self, stage: int) -> tuple[float, int]:
        """(sum of the event-timed kernel intervals of a YK_STAGE_* since the last query, number of intervals)."""
        ms, n = C.c_float(), C.c_int()
        _chk(self._h, self._L.yk_stage_ms(self._h, stage, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def export_capacity(self) -> int:
        return int(self._L.yk_export_capacity(self._h))

    def export_tile_maps(self, dev_buffer) -> np.ndarray:
        """dev_buffer: torch uint8 cuda tensor of >= export_capacity() bytes. Returns the 15 section sizes."""
        sizes = np.zeros(15, dtype=np.uint64)
        # hand-over fence (see set_image): whatever torch still has queued on the buffer (its allocation fill, an earlier consumer) must be
        # done before the handle's own stream writes into it -- the two streams are not ordered against each other
        import torch
        torch.cuda.current_stream(dev_buffer.device).synchronize()
        _chk(self._h, self._L.yk_export_tile_maps(self._h, C.c_void_p(dev_buffer.data_ptr()), dev_buffer.numel(), sizes.ctypes.data))
        return sizes

    def export_tile_maps_async(self, dev_buffer, dev_meta16, consumer_stream: int = 0) -> None:
        """No host synchronisation: dev_meta16 (torch int64[16] cuda tensor) receives {total bytes, sizes[0..14]}; work queued
        afterwards on `consumer_stream` (a hipStream_t of the same runtime, 0 = null stream) sees buffer and table complete.
        The caller orders EARLIER work on the two buffers before the handle's stream itself (yk_stream_wait_for, or buffers that are idle)."""
        _chk(self._h, self._L.yk_export_tile_maps_async(self._h, C.c_void_p(dev_buffer.data_ptr()), dev_buffer.numel(),
                                                      C.c_void_p(dev_meta16.data_ptr()), C.c_void_p(consumer_stream)))

    def export_tile_maps_framed(self, dev_buffer, consumer_stream: int | None = 0) -> None:
        """The form the gather moves (yk_export_tile_maps_framed): dev_buffer[0:128] = header {payload bytes, sizes[0..14]}, the sections
        behind it; no host synchronisation.  Work queued afterwards on `consumer_stream` (0 = null stream; None = no hand-over) sees the
        buffer.  The caller orders EARLIER work on the buffer before the handle's stream (idle buffers, or yk_stream_wait_for)."""
        cs = C.c_void_p(-1 & 0xFFFFFFFFFFFFFFFF) if consumer_stream is None else C.c_void_p(consumer_stream)
        _chk(self._h, self._L.yk_export_tile_maps_framed(self._h, C.c_void_p(dev_buffer.data_ptr()), dev_buffer.numel(), cs))

    def kernel_ms(self) -> dict:
        e, a, p = C.c_float(), C.c_float(), C.c_float()
        _chk(self._h, self._L.yk_last_kernel_ms(self._h, C.byref(e), C.byref(a), C.byref(p)))
        return {"encode": e.value, "alpha": a.value, "pack": p.value}
