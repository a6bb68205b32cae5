"""Deterministic synthetic input "YAIK-synth v1" (SURVEY.md §8d).

Planar int32 planes with values in [0, 255], laid out exactly like the reference's ``Plane``
(row-major, ``idx = x + y*w``; encoder/framework.h:82).  Used by the tests, the golden-vector
generator and bench.py so that the CPU baseline and the GPU path see identical pixels.
"""
from __future__ import annotations

import numpy as np

_A = 1664525
_C = 1013904223
_MASK = 0xFFFFFFFF


def _lcg_stream(n: int, seed: int) -> np.ndarray:
    """s_k = LCG^(k+1)(seed) for k in [0, n) as uint32, via jump-ahead tables (no Python loop per pixel)."""
    chunk = 1 << 16
    # per-offset multipliers/increments inside a chunk: s_{j} = A[j]*s0 + Cc[j]  (mod 2^32), j = 1..chunk
    A = np.empty(chunk, dtype=np.uint64)
    Cc = np.empty(chunk, dtype=np.uint64)
    A[0] = _A
    Cc[0] = _C
    filled = 1
    while filled < chunk:
        take = min(filled, chunk - filled)
        aL = A[filled - 1]
        cL = Cc[filled - 1]
        A[filled:filled + take] = (A[:take] * aL) & _MASK
        Cc[filled:filled + take] = (Cc[:take] * aL + cL) & _MASK
        filled += take
    out = np.empty(n, dtype=np.uint32)
    s0 = np.uint64(seed & _MASK)
    pos = 0
    while pos < n:
        m = min(chunk, n - pos)
        vals = (A[:m] * s0 + Cc[:m]) & _MASK
        out[pos:pos + m] = vals.astype(np.uint32)
        s0 = vals[m - 1]
        pos += m
    return out


def synth_planes(w: int, h: int | None = None, n_planes: int = 4, seed: int = 12345) -> np.ndarray:
    """Return an int32 array [n_planes, h, w] (R, G, B[, A]).  ``w`` is the ramp/class scale (square rule of §8d)."""
    if h is None:
        h = w
    W = w
    s = _lcg_stream(w * h, seed).reshape(h, w)
    y, x = np.meshgrid(np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64), indexing="ij")
    k = ((x >> 6) + (y >> 6)) & 3
    r = (255 * x) // W
    g = np.minimum((255 * y) // W, 255)
    b = np.minimum((255 * (x + y)) // (2 * W), 255)
    s64 = s.astype(np.int64)
    n2 = np.stack([(s64 >> 8) & 7, (s64 >> 12) & 7, (s64 >> 16) & 7])
    n3 = np.stack([(s64 >> 8) & 255, (s64 >> 16) & 255, (s64 >> 24) & 255])
    base = np.stack([r, g, b])
    rgb = np.where(k == 2, (base + n2) % 256, base)
    rgb = np.where(k == 3, n3, rgb)
    planes = [rgb[0], rgb[1], rgb[2]]
    if n_planes == 4:
        frame = (x < W // 8) | (x >= W - W // 16) | (y < W // 16) | (y >= W - W // 8)
        holes = (((x >> 7) + (y >> 7)) % 5) == 0
        planes.append(np.where(frame | holes, 0, 255))
    return np.ascontiguousarray(np.stack(planes).astype(np.int32))


def synth_planes_torch(w: int, h: int | None = None, n_planes: int = 4, seed: int = 12345, device="cuda", row0: int = 0, rows: int | None = None):
    """Same image as synth_planes, generated on `device` with torch (int32 tensor [n, rows, w]).
    The LCG is evaluated in closed form per pixel (affine-map power by squaring, mod 2^32), so any band of rows
    [row0, row0 + rows) of the w x h image can be generated on its own (row stripes of one large image)."""
    import torch
    if h is None:
        h = w
    if rows is None:
        rows = h - row0
    W = w
    n = w * rows
    M = 0xFFFFFFFF
    k = torch.arange(1, n + 1, dtype=torch.int64, device=device) + row0 * w          # pixel i uses LCG^(i+1)(seed)
    A = torch.ones(n, dtype=torch.int64, device=device)
    Cc = torch.zeros(n, dtype=torch.int64, device=device)
    pa, pc = _A, _C
    for bit in range(max(1, int(w * h).bit_length())):
        sel = ((k >> bit) & 1).bool()
        A = torch.where(sel, (A * pa) & M, A)
        Cc = torch.where(sel, (Cc * pa + pc) & M, Cc)
        pc = (pa * pc + pc) & M
        pa = (pa * pa) & M
    s = ((A * (seed & M) + Cc) & M).reshape(rows, w)
    del A, Cc, k
    y = torch.arange(row0, row0 + rows, dtype=torch.int64, device=device).reshape(rows, 1).expand(rows, w)
    x = torch.arange(w, dtype=torch.int64, device=device).reshape(1, w).expand(rows, w)
    kk = ((x >> 6) + (y >> 6)) & 3
    base = [(255 * x) // W, torch.clamp((255 * y) // W, max=255), torch.clamp((255 * (x + y)) // (2 * W), max=255)]
    n2 = [(s >> 8) & 7, (s >> 12) & 7, (s >> 16) & 7]
    n3 = [(s >> 8) & 255, (s >> 16) & 255, (s >> 24) & 255]
    out = torch.empty((n_planes, rows, w), dtype=torch.int32, device=device)
    for c in range(3):
        v = torch.where(kk == 2, (base[c] + n2[c]) % 256, base[c])
        v = torch.where(kk == 3, n3[c], v)
        out[c] = v.to(torch.int32)
    if n_planes == 4:
        frame = (x < W // 8) | (x >= W - W // 16) | (y < W // 16) | (y >= W - W // 8)
        holes = (((x >> 7) + (y >> 7)) % 5) == 0
        out[3] = torch.where(frame | holes, 0, 255).to(torch.int32)
    return out


# ---- synthetic 3-D LUT bank: the reference's own bank (22 'Bank3D//*.lut' files) is not in its repository -------------------
def bank_patterns(n_patterns: int = 6) -> list:
    """Point-cloud patterns in the 64^3 cube (uint8 [count, 3], 6-bit r, g, b): the content of `Load3DPattern` files (EncoderContext.cpp:7851)."""
    i = np.arange(64, dtype=np.float64)
    t = i / 63.0
    pats = [
        np.stack([i, i, i], 1),                                                    # the diagonal: colours between two end points
        np.stack([i, 63 * t * t, i], 1),                                           # one channel lags
        np.stack([i, 63 * np.sqrt(t), 63 * t * t], 1),                             # one leads, one lags
        np.stack([i, np.minimum(2 * i, 63), np.maximum(2 * i - 63, 0)], 1),        # a bent path through a cube edge
        np.stack([63 * (1 - np.cos(np.pi * t)) / 2, i, 63 - i], 1),                # S-curve against a falling channel
        np.stack([i[:40] * 63 / 39, 63 - i[:40] * 63 / 39, i[:40] * 63 / 39 * 0.5], 1),   # 40 points only
    ]
    return [np.clip(np.floor(p + 0.5), 0, 63).astype(np.uint8) for p in pats[:n_patterns]]


def bank_bytes(patterns) -> bytes:
    out = bytearray()
    for p in patterns:
        out.append(len(p))
        out += p[:, 0].tobytes() + p[:, 1].tobytes() + p[:, 2].tobytes()
    return bytes(out)


