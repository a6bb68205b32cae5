"""TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liboracle.so (the plain-C CPU restatement).

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

PASSES = [(4, 4), (4, 3), (3, 4), (3, 3), (3, 2), (2, 3), (2, 2)]   # EncoderContext.cpp:9057-9093


def build() -> None:
    subprocess.run(["make", "-C", HERE, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    lib.yko_enc_create.restype = vp
    lib.yko_enc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.yko_enc_destroy.argtypes = [vp]
    lib.yko_mip_prefilter.argtypes = [vp]
    lib.yko_get_bounds.argtypes = [vp, vp]
    for name in ("yko_mip_bitmap", "yko_last_bitmap", "yko_last_rgb_stream", "yko_1d_pix_stream", "yko_1d_type_stream",
                 "yko_last_tile_defs"):
        getattr(lib, name).restype = vp
        getattr(lib, name).argtypes = [vp, ip]
    lib.yko_last_nibbles.restype = vp
    lib.yko_last_nibbles.argtypes = [vp, ip, ip]
    lib.yko_fitting_quad_smooth.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    for name in ("yko_smooth_map", "yko_mipmap_mask"):
        getattr(lib, name).restype = vp
        getattr(lib, name).argtypes = [vp]
    for name in ("yko_map_smooth_tile", "yko_preview"):
        getattr(lib, name).restype = vp
        getattr(lib, name).argtypes = [vp, C.c_int]
    lib.yko_dynamic_tile_encode.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.yko_dynamic_tile_compressor.argtypes = [vp, C.c_int, vp]
    lib.yko_build_table.argtypes = [C.c_int, C.c_int, vp]
    lib.yko_curve_constants.argtypes = [vp]
    lib.yko_palette_remap.argtypes = [vp, C.c_int, C.c_int]
    lib.yko_palette_compress.argtypes = [vp, vp, C.c_int]
    lib.yko_last_palette.restype = vp
    lib.yko_last_palette.argtypes = [vp, ip]
    lib.yko_palette_decompress.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]
    lib.yko_dec_create.restype = vp
    lib.yko_dec_create.argtypes = [C.c_int, C.c_int]
    lib.yko_dec_destroy.argtypes = [vp]
    lib.yko_dec_gradient.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int]
    lib.yko_dec_gradient_planes.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int]
    lib.yko_dec_split_masks.argtypes = [vp]
    lib.yko_dec_1d.argtypes = [vp, C.c_int, vp, ip, vp, ip, C.c_int]
    lib.yko_dec_mask.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.yko_image_builder.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int]
    for name in ("yko_dec_planes", "yko_dec_tile4x4", "yko_dec_map_rgb", "yko_dec_map_rgb_mask"):
        getattr(lib, name).restype = vp
        getattr(lib, name).argtypes = [vp, ip]
    # (f)4 3-D LUT search
    lib.yko_lut_load.argtypes = [vp, vp, vp, vp, C.c_int]
    lib.yko_lut_count.argtypes = [vp]
    for name in ("yko_lut_factors", "yko_lut_distance_field", "yko_lut_preview"):
        getattr(lib, name).restype = vp
        getattr(lib, name).argtypes = [vp, C.c_int]
    lib.yko_lut_positions.restype = vp
    lib.yko_lut_positions.argtypes = [vp, C.c_int, C.c_int]
    lib.yko_lut_start.argtypes = [vp]
    lib.yko_lut_search.argtypes = [vp, C.c_int, C.c_int]
    for name in ("yko_lut_tile_types", "yko_lut_colors"):
        getattr(lib, name).restype = vp
        getattr(lib, name).argtypes = [vp, ip]
    for name in ("yko_lut_indices", "yko_lut_map"):
        getattr(lib, name).restype = vp
        getattr(lib, name).argtypes = [vp, C.c_int, ip]
    lib.yko_lut_file.argtypes = [vp, vp, C.c_int]
    lib.yko_dec_lut3d.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp]
    return lib


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _arr(ptr, n, dtype=np.uint8) -> np.ndarray:
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


class OracleEncoder:
    """Mirror of the EncoderContext hot-path operator surface, on the CPU oracle."""

    def __init__(self, planes: np.ndarray):
        self.planes = np.ascontiguousarray(planes, dtype=np.int32)
        self.n, self.h, self.w = self.planes.shape
        ptrs = (C.c_void_p * 4)(*[self.planes[i].ctypes.data if i < self.n else None for i in range(4)])
        self._e = lib().yko_enc_create(self.w, self.h, self.n, ptrs)
        if not self._e:
            raise ValueError("bad geometry")

    def __del__(self):
        if getattr(self, "_e", None):
            lib().yko_enc_destroy(self._e)
            self._e = None

    def mip_prefilter(self) -> dict:
        rc = lib().yko_mip_prefilter(self._e)
        if rc < 0:
            raise ValueError("MipPrefilter recursion only defined for square power-of-two images")
        b = np.zeros(10, dtype=np.int32)
        lib().yko_get_bounds(self._e, b.ctypes.data)
        n = C.c_int()
        p = lib().yko_mip_bitmap(self._e, C.byref(n))
        return {"has_chunk": bool(rc), "bounds": b[:4].copy(), "tile_size": int(b[4]), "remaining": int(b[5]),
                "tile_bbox": b[6:10].copy(), "bitmap": _arr(p, n.value)}

    def bounds(self) -> np.ndarray:
        b = np.zeros(10, dtype=np.int32)
        lib().yko_get_bounds(self._e, b.ctypes.data)
        return b[:4].copy()

    def fitting_quad_smooth(self, sx: int, sy: int, reject_factor: int = 3, plane_bit: int = 7):
        cnt = lib().yko_fitting_quad_smooth(self._e, reject_factor, plane_bit, sx, sy)
        n = C.c_int()
        p = lib().yko_last_bitmap(self._e, C.byref(n)); bitmap = _arr(p, n.value)
        p = lib().yko_last_rgb_stream(self._e, C.byref(n)); rgb = _arr(p, n.value)
        return cnt, bitmap, rgb

    def palette_compress(self, rgb: np.ndarray) -> np.ndarray:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        n = lib().yko_palette_compress(self._e, rgb.ctypes.data, rgb.size)
        if n < 0:
            raise RuntimeError("PaletteCompressor overflow")
        c = C.c_int()
        p = lib().yko_last_palette(self._e, C.byref(c))
        return _arr(p, c.value)

    # ---- (f)4 3-D LUT tile search (Load3DPattern / Correlation3DSearch / computeValues3D) -------------------------------
    def lut_load(self, pattern: np.ndarray) -> int:
        """pattern: uint8 [count, 3], 6-bit coordinates (the contents of one Bank3D .lut file)."""
        p = np.ascontiguousarray(pattern, dtype=np.uint8)
        r, g, b = (np.ascontiguousarray(p[:, c]) for c in range(3))
        k = lib().yko_lut_load(self._e, r.ctypes.data, g.ctypes.data, b.ctypes.data, len(p))
        if k < 0:
            raise ValueError("pattern rejected (1..64 points, at most 64 patterns)")
        return k

    def lut_tables(self, k: int):
        """(factors int16 [4, 3, 64], distanceField int32 [64^3], positions uint8 [4, 64^3]) of pattern k; axis 0 = 6, 5, 4, 3 bits."""
        L = lib()
        fac = _arr(L.yko_lut_factors(self._e, k), 4 * 3 * 64, np.int16).reshape(4, 3, 64)
        dist = _arr(L.yko_lut_distance_field(self._e, k), 64 ** 3, np.int32)
        pos = np.stack([_arr(L.yko_lut_positions(self._e, k, s), 64 ** 3) for s in range(4)])
        return fac, dist, pos

    def lut_file(self) -> np.ndarray:
        """The decoder's 'LUL0' LUT file for the loaded bank (what RegisterAndCreate3DLut writes to LutFile.lut)."""
        n = lib().yko_lut_file(self._e, None, 0)
        out = np.zeros(n, np.uint8)
        lib().yko_lut_file(self._e, out.ctypes.data, n)
        return out

    def lut_start(self) -> None:
        lib().yko_lut_start(self._e)

    def lut_search(self, sx: int, sy: int) -> int:
        return lib().yko_lut_search(self._e, sx, sy)

    def lut_streams(self) -> dict:
        L, n = lib(), C.c_int()
        out = {}
        p = L.yko_lut_tile_types(self._e, C.byref(n)); out["tileType"] = _arr(p, n.value, np.uint16)
        p = L.yko_lut_colors(self._e, C.byref(n)); out["color"] = _arr(p, n.value)
        for bits in (3, 4, 5, 6):
            p = L.yko_lut_indices(self._e, bits, C.byref(n)); out[f"idx{bits}"] = _arr(p, n.value)
        for k in range(6):
            p = L.yko_lut_map(self._e, k, C.byref(n)); out[f"map{k}"] = _arr(p, n.value)
        out["preview"] = np.stack([_arr(L.yko_lut_preview(self._e, c), self.w * self.h, np.int32).reshape(self.h, self.w) for c in range(3)])
        return out

    def state(self, name: str, plane: int = 0) -> np.ndarray:
        n = self.w * self.h
        L = lib()
        if name == "smoothMap":
            return _arr(L.yko_smooth_map(self._e), n).reshape(self.h, self.w)
        if name == "mipmapMask":
            return _arr(L.yko_mipmap_mask(self._e), n).reshape(self.h, self.w)
        if name == "mapSmoothTile":
            return _arr(L.yko_map_smooth_tile(self._e, plane), n).reshape(self.h, self.w)
        if name == "preview":
            return _arr(L.yko_preview(self._e, plane), n, np.int32).reshape(self.h, self.w)
        raise KeyError(name)

    def dynamic_tile_encode(self, plane: int, mode3bit_only: bool, fill: int = -1):
        dst = np.full((self.h, self.w), fill, dtype=np.int32)
        lib().yko_dynamic_tile_encode(self._e, int(mode3bit_only), plane, dst.ctypes.data)
        n, nn = C.c_int(), C.c_int()
        p = lib().yko_last_tile_defs(self._e, C.byref(n)); defs = _arr(p, n.value, np.uint16)
        p = lib().yko_last_nibbles(self._e, C.byref(n), C.byref(nn)); nib = _arr(p, n.value)
        return defs, nib, nn.value, dst

    def dynamic_tile_compressor(self, plane: int):
        dbg = np.zeros((self.h, self.w), dtype=np.int32)
        tiles = lib().yko_dynamic_tile_compressor(self._e, plane, dbg.ctypes.data)
        return tiles, dbg

    def streams_1d(self):
        n = C.c_int()
        p = lib().yko_1d_pix_stream(self._e, C.byref(n)); pix = _arr(p, n.value)
        p = lib().yko_1d_type_stream(self._e, C.byref(n)); typ = _arr(p, n.value)
        return pix, typ


class OracleDecoder:
    def __init__(self, w: int, h: int):
        self.w, self.h = w, h
        self._d = lib().yko_dec_create(w, h)

    def __del__(self):
        if getattr(self, "_d", None):
            lib().yko_dec_destroy(self._d)
            self._d = None

    def gradient(self, sx: int, sy: int, bitmap: np.ndarray, rgb_dq: np.ndarray) -> int:
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
        rgb_dq = np.ascontiguousarray(rgb_dq, dtype=np.uint8)
        return lib().yko_dec_gradient(self._d, sx, sy, bitmap.ctypes.data, bitmap.size,
                                      rgb_dq.ctypes.data if rgb_dq.size else None, rgb_dq.size)

    def gradient_planes(self, plane_bit: int, bitmap: np.ndarray, rgb_dq: np.ndarray, consistent_marks: bool = False) -> int:
        """DecompressGradient4x4 with planeBit 1..6 (masks must have been split, like the decoder does for such a chunk).
        consistent_marks=False reproduces the reference's tile4x4Mask marking defects (pinned); True marks every present plane."""
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
        rgb_dq = np.ascontiguousarray(rgb_dq, dtype=np.uint8)
        return lib().yko_dec_gradient_planes(self._d, plane_bit, int(consistent_marks), bitmap.ctypes.data, bitmap.size,
                                             rgb_dq.ctypes.data if rgb_dq.size else None, rgb_dq.size)

    def lut3d(self, lut_file: np.ndarray, maps, tiles: np.ndarray, colors_dq: np.ndarray, idx) -> np.ndarray:
        """Tile3D_16x8 .. 4x4 on the streams of a '3DTL' chunk (maps: the six tile maps in chunk order; colors_dq: after
        PaletteFullRangeRemapping; idx: the 3/4/5/6-bit index streams as stored, x 3).  Returns the bytes consumed per stream."""
        pad = lambda a, n, dt=np.uint8: np.ascontiguousarray(np.concatenate([np.asarray(a, dt).ravel(), np.zeros(n, dt)]))
        lf = np.ascontiguousarray(lut_file, np.uint8)
        mp = [pad(m, 64) for m in maps]
        ix = [pad(i, 256) for i in idx]
        t = pad(tiles, 32, np.uint16); cdq = pad(colors_dq, 64)
        mptr = (C.c_void_p * 6)(*[m.ctypes.data for m in mp])
        iptr = (C.c_void_p * 4)(*[i.ctypes.data for i in ix])
        used = np.zeros(6, np.int32)
        rc = lib().yko_dec_lut3d(self._d, lf.ctypes.data, lf.size, mptr, t.ctypes.data, int(np.asarray(tiles).size), cdq.ctypes.data, iptr, used.ctypes.data)
        if rc:
            raise ValueError("invalid LUT file")
        return used

    def split_masks(self):
        lib().yko_dec_split_masks(self._d)

    def decode_1d(self, typ: np.ndarray, pix: np.ndarray, compression_range: int = 15):
        typ = np.ascontiguousarray(np.concatenate([typ, np.zeros(64, np.uint8)]))
        pix = np.ascontiguousarray(np.concatenate([pix, np.zeros(64, np.uint8)]))
        tp, pp = C.c_int(0), C.c_int(0)
        for p in range(3):
            lib().yko_dec_1d(self._d, p, typ.ctypes.data, C.byref(tp), pix.ctypes.data, C.byref(pp), compression_range)
        return tp.value, pp.value

    def planes(self) -> np.ndarray:
        n = C.c_int()
        p = lib().yko_dec_planes(self._d, C.byref(n))
        return _arr(p, n.value * 3).reshape(3, n.value)

    def tile4x4(self, all_planes: bool = False) -> np.ndarray:
        n = C.c_int()
        p = lib().yko_dec_tile4x4(self._d, C.byref(n))
        return _arr(p, n.value * (3 if all_planes else 1))

    def map_rgb(self) -> np.ndarray:
        n = C.c_int()
        p = lib().yko_dec_map_rgb(self._d, C.byref(n))
        return _arr(p, n.value)

    def map_rgb_mask(self, all_planes: bool = False) -> np.ndarray:
        n = C.c_int()
        p = lib().yko_dec_map_rgb_mask(self._d, C.byref(n))
        return _arr(p, n.value * (3 if all_planes else 1))


def yko_compress_f(values: np.ndarray, rate: int) -> np.ndarray:
    """CompressF (EncoderContext.cpp:3191) per byte: (v * rate + 127) / 255."""
    v = np.ascontiguousarray(values, dtype=np.uint8).astype(np.int64)
    return ((v * rate + 127) // 255).astype(np.uint8)


def palette_remap(stream: np.ndarray, original_range: int = 250) -> np.ndarray:
    out = np.ascontiguousarray(stream, dtype=np.uint8).copy()
    if out.size:
        lib().yko_palette_remap(out.ctypes.data, out.size, original_range)
    return out


def palette_decompress(pal: np.ndarray, out_size: int, color_compression: int = 250) -> np.ndarray:
    buf = np.concatenate([np.ascontiguousarray(pal, dtype=np.uint8), np.zeros(128 * 3 + 8, np.uint8)])
    out = np.zeros(out_size, dtype=np.uint8)
    ok = lib().yko_palette_decompress(buf.ctypes.data, int(pal.size), out.ctypes.data, out_size, color_compression)
    if not ok:
        raise RuntimeError("PaletteDecompressor rejected the stream")
    return out


def dec_mask(bits: np.ndarray, bw: int, bh: int) -> np.ndarray:
    """Decompress1BitTiled (decoder/YAIK_Mipmap.cpp:23-154): 1 bit per 16x16 tile of the bw x bh tile box -> swizzled 1 bit / pixel."""
    src = np.concatenate([np.ascontiguousarray(bits, dtype=np.uint8), np.zeros(8, np.uint8)])
    out = np.zeros((bw * bh * 256) // 8, dtype=np.uint8)
    lib().yko_dec_mask(src.ctypes.data, bw, bh, out.ctypes.data)
    return out


def image_builder(planes_tiled: np.ndarray, w: int, h: int, stride: int, alpha: np.ndarray = None, fill: int = 0xA5) -> np.ndarray:
    """internal_imageBuilderFunc (decoder/YAIK_DefaultCallback.cpp:24-191) -> [h, stride] bytes; bytes the reference does not
    write keep `fill`.  alpha (h x w u8, linear) selects the reference's RGBA branch, reproduced with its defects."""
    pl = np.ascontiguousarray(planes_tiled, dtype=np.uint8).reshape(3, -1)
    out = np.full((h, stride), fill, dtype=np.uint8)
    a = None if alpha is None else np.ascontiguousarray(alpha, dtype=np.uint8)
    rc = lib().yko_image_builder(pl.ctypes.data, pl.shape[1], w, h, None if a is None else a.ctypes.data, 0 if a is None else a.shape[1],
                                 out.ctypes.data, stride)
    if rc != 0:
        raise ValueError("image_builder: width / height must be multiples of 8")
    return out


def detile(plane_tiled: np.ndarray, w: int, h: int) -> np.ndarray:
    """8x8-tiled u8 plane (include/YAIK.h:205-224) -> row-major [h, w]."""
    tw, th = (w + 7) // 8, (h + 7) // 8
    return plane_tiled.reshape(th, tw, 8, 8).transpose(0, 2, 1, 3).reshape(th * 8, tw * 8)[:h, :w]
