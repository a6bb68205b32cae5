"""Python plumbing over the decode half of the C-ABI (include/yaik_hip.h): the loops behind
YAIK_DecodeImage's chunk switch (decoder/YAIK_API.cpp:731-1303), executed on the GPU.

Method names follow the reference: DecompressGradient*, Decompress1D, Decompress1BitTiled.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import YaikError, lib
from .encoder import _chk


class HipTileDecoder:
    def __init__(self, device: int = 0):
        h = C.c_void_p()
        rc = lib().yk_create(device, C.byref(h))
        if rc != 0:
            raise YaikError(f"yk_create failed ({rc}): no usable HIP device -- the product path has no CPU fallback")
        self._h = h
        self.w = self.h = 0

    def close(self):
        if getattr(self, "_h", None):
            lib().yk_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def begin(self, w: int, h: int):
        self.w, self.h = w, h
        _chk(self._h, lib().yk_decode_begin(self._h, w, h))

    def decompress_gradient(self, sx: int, sy: int, bitmap: np.ndarray, rgb_dq: np.ndarray):
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
        rgb_dq = np.ascontiguousarray(rgb_dq, dtype=np.uint8)
        _chk(self._h, lib().yk_decode_gradient(self._h, sx, sy, bitmap.ctypes.data, bitmap.size,
                                               rgb_dq.ctypes.data if rgb_dq.size else None, rgb_dq.size))

    def encoder_streams(self, enc) -> list:
        """The device-resident streams of an encoder handle on the same device, as the argument lists of yk_decode_gradient_device (one per
        gradient pass with accepted tiles) and yk_decode_1d_device: tile bitmaps, corner streams and the 1-D streams where the encoder left
        them in HBM.  Runs the encoder's corner and 1-D stages if they have not run, and fences the encoder once (for the lengths)."""
        shapes = [(4, 4), (4, 3), (3, 4), (3, 3), (3, 2), (2, 3), (2, 2)]
        counts = enc.gradient_counts()
        calls = []
        for i, (sx, sy) in enumerate(shapes):
            dev, n = C.c_void_p(), C.c_size_t()
            _chk(enc._h, enc._L.yk_gradient_corners_device(enc._h, i, C.byref(dev), C.byref(n)))
            if counts[i]:
                calls.append(("g", sx, sy, enc._L.yk_gradient_bitmap_device(enc._h, i), enc._L.yk_gradient_bitmap_bytes(enc._h, i), dev, n.value))
        _chk(enc._h, enc._L.yk_range1d_encode(enc._h))
        pix, npx, typ, nty = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        _chk(enc._h, enc._L.yk_range1d_streams_device(enc._h, C.byref(pix), C.byref(npx), C.byref(typ), C.byref(nty)))
        calls.append(("1", typ, nty.value, pix, npx.value))
        enc.synchronize()
        return calls

    def decode_streams(self, calls: list, sync: bool = True, per_pass: bool = False) -> None:
        """All gradient chunks + the 1-D chunk from device-resident streams (encoder_streams): the corner streams are remapped like
        PaletteFullRangeRemapping(250) on the way in; no PCIe hop, no host synchronisation between the passes.  The gradient chunks go through
        ONE yk_decode_gradient_all_device call (per_pass=True: one yk_decode_gradient_device call per chunk, same result)."""
        L = lib()
        g = [c for c in calls if c[0] == "g"]
        if per_pass:
            for c in g:
                _chk(self._h, L.yk_decode_gradient_device(self._h, c[1], c[2], c[3], c[4], c[5], c[6], 250))
        elif g:
            n = len(g)
            ptr = lambda v: v.value if isinstance(v, C.c_void_p) else v
            sx, sy = (C.c_int * n)(*[c[1] for c in g]), (C.c_int * n)(*[c[2] for c in g])
            bm, nb = (C.c_void_p * n)(*[ptr(c[3]) for c in g]), (C.c_size_t * n)(*[c[4] for c in g])
            rgb, nr = (C.c_void_p * n)(*[ptr(c[5]) for c in g]), (C.c_size_t * n)(*[c[6] for c in g])
            _chk(self._h, L.yk_decode_gradient_all_device(self._h, n, sx, sy, bm, nb, rgb, nr, 250))
        for c in calls:
            if c[0] == "1":
                _chk(self._h, L.yk_decode_1d_device(self._h, c[1], c[2], c[3], c[4], 15))
        if sync:
            _chk(self._h, L.yk_synchronize(self._h))

    def decode_from_encoder(self, enc, sync: bool = True, per_pass: bool = False) -> None:
        self.decode_streams(self.encoder_streams(enc), sync, per_pass)

    def decompress_gradient_planes(self, plane_bit: int, bitmap: np.ndarray, rgb_dq: np.ndarray, consistent_marks: bool = False):
        """DecompressGradient4x4 with planeBit 1..6; consistent_marks=False leaves tile4x4Mask as the reference's loops do (defects included)."""
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
        rgb_dq = np.ascontiguousarray(rgb_dq, dtype=np.uint8)
        _chk(self._h, lib().yk_decode_gradient_planes(self._h, plane_bit, int(consistent_marks), bitmap.ctypes.data, bitmap.size,
                                                      rgb_dq.ctypes.data if rgb_dq.size else None, rgb_dq.size))

    def assign_lut(self, lut_file: np.ndarray) -> None:
        """YAIK_AssignLUT: the decoder's 3-D LUT file ('LUL0')."""
        lf = np.ascontiguousarray(lut_file, dtype=np.uint8)
        _chk(self._h, lib().yk_decode_assign_lut(self._h, lf.ctypes.data, lf.size))

    def decompress_lut3d(self, maps, tiles: np.ndarray, colors_dq: np.ndarray, idx) -> np.ndarray:
        """The '3DTL' chunk: Tile3D_16x8 .. Tile3D_4x4 on its streams (see yk_decode_lut3d).  Returns the bytes consumed per stream."""
        mp = [np.ascontiguousarray(m, dtype=np.uint8) for m in maps]
        ix = [np.ascontiguousarray(i, dtype=np.uint8) for i in idx]
        t = np.ascontiguousarray(tiles, dtype=np.uint16); cdq = np.ascontiguousarray(colors_dq, dtype=np.uint8)
        mptr = (C.c_void_p * 6)(*[m.ctypes.data if m.size else None for m in mp])
        msz = (C.c_size_t * 6)(*[m.size for m in mp])
        iptr = (C.c_void_p * 4)(*[i.ctypes.data if i.size else None for i in ix])
        isz = (C.c_size_t * 4)(*[i.size for i in ix])
        used = (C.c_size_t * 6)()
        _chk(self._h, lib().yk_decode_lut3d(self._h, mptr, msz, t.ctypes.data if t.size else None, t.size, cdq.ctypes.data if cdq.size else None, iptr, isz, used))
        return np.array(list(used), dtype=np.int64)

    def decompress_1d(self, type_stream: np.ndarray, pix_stream: np.ndarray, compression_range: int = 15):
        t = np.ascontiguousarray(type_stream, dtype=np.uint8)
        p = np.ascontiguousarray(pix_stream, dtype=np.uint8)
        if t.size == 0 or p.size == 0:
            return
        _chk(self._h, lib().yk_decode_1d(self._h, t.ctypes.data, t.size, p.ctypes.data, p.size, compression_range))

    def decompress_1bit_tiled(self, bits: np.ndarray, bw: int, bh: int) -> np.ndarray:
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        out = np.zeros(bw * bh * 32, dtype=np.uint8)
        _chk(self._h, lib().yk_decode_mask(self._h, bits.ctypes.data, bw, bh, out.ctypes.data, out.size))
        return out

    def planes(self) -> np.ndarray:
        n = (self.w // 8) * (self.h // 8) * 64
        out = np.zeros((3, n), dtype=np.uint8)
        _chk(self._h, lib().yk_decode_planes(self._h, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, n))
        return out

    def image(self, alpha: np.ndarray | None = None, stride: int | None = None, fill: int = 0, reference_rgba: bool = False) -> np.ndarray:
        """internal_imageBuilderFunc: interleaved RGB ([h, stride] bytes, 3 B/pixel) or RGBA when an alpha plane is given.
        Bytes of a row beyond the pixels keep `fill` (the call never writes them).  reference_rgba selects the reference's own
        RGBA branch, defects included (yk_decode_output_reference_rgba)."""
        bpp = 4 if alpha is not None else 3
        stride = stride or self.w * bpp
        out = np.full((self.h, stride), fill, dtype=np.uint8)
        a = np.ascontiguousarray(alpha, dtype=np.uint8) if alpha is not None else None
        fn = lib().yk_decode_output_reference_rgba if reference_rgba else lib().yk_decode_output
        _chk(self._h, fn(self._h, out.ctypes.data, stride, a.ctypes.data if a is not None else None, a.shape[1] if a is not None else 0))
        return out

    def image_into(self, out: np.ndarray) -> None:
        """RGB rows into a caller-owned [h, stride] uint8 array (no allocation per call)."""
        _chk(self._h, lib().yk_decode_output(self._h, out.ctypes.data, out.shape[1], None, 0))

    def synchronize(self):
        _chk(self._h, lib().yk_synchronize(self._h))

    def stage_ms(self, stage: int) -> tuple[float, int]:
        ms, n = C.c_float(), C.c_int()
        _chk(self._h, lib().yk_stage_ms(self._h, stage, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def tile4x4(self, all_planes: bool = False) -> np.ndarray:
        n = ((((self.w + 15) >> 4) << 2) * (((self.h + 7) >> 3) << 1)) >> 3
        out = np.zeros(n * (3 if all_planes else 1), dtype=np.uint8)
        fn = lib().yk_decode_tile4x4_planes if all_planes else lib().yk_decode_tile4x4
        _chk(self._h, fn(self._h, out.ctypes.data, out.size))
        return out
