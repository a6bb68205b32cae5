"""A SYNTHETIC 3-D LUT bank for (f)4, in the file format the reference's Load3DPattern reads (EncoderContext.cpp:7851-7867):
u8 count, then count x r, count x g, count x b, every value 6 bits (a point of the 64^3 cube the tile colours are normalised into).
The reference's own bank (22 'Bank3D//*.lut' files, :7796-7819) is not in its repository; this generator lets the reference, the oracle and
the HIP path run the same search.  Patterns hold at most 64 points: beyond that Load3DPattern reduces the array to 64 entries but still
hands Set3DPointCloud the original count (:7907-7917), which reads and writes past its 64-entry tables."""
import numpy as np


from yaik_amd.synth import bank_bytes, bank_patterns  # noqa: E402,F401  (the generator lives with the other synthetic inputs)


def lut_image(w: int, h: int, patterns=None, seed: int = 3) -> np.ndarray:
    """int32 planes [3, h, w]: 16x16 blocks whose colours run along a bank pattern between two random end colours with a non-bilinear
    parameter field (the gradient passes reject them, the 3-D LUT search should take them), mixed with ramp blocks (gradient tiles),
    noise blocks (nothing matches) and blocks that are flat in one channel."""
    patterns = patterns or bank_patterns()
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    out = np.zeros((3, h, w), dtype=np.int64)
    ramp = np.stack([(x * 255) // w, (y * 255) // h, ((x + y) * 255) // (w + h)])
    for by in range(0, h, 16):
        for bx in range(0, w, 16):
            sl = (slice(None), slice(by, min(by + 16, h)), slice(bx, min(bx + 16, w)))
            kind = rng.integers(0, 8)
            yy, xx = y[sl[1:]] - by, x[sl[1:]] - bx
            if kind == 0:
                out[sl] = ramp[sl]
            elif kind == 1:
                out[sl] = rng.integers(0, 256, out[sl].shape)
            else:
                e = int(rng.integers(0, len(patterns)))
                pts = patterns[e].astype(np.float64) / 63.0
                c0 = rng.integers(0, 200, 3).astype(np.float64)
                c1 = c0 + rng.integers(20, 56, 3)
                if kind == 2:
                    c1[int(rng.integers(0, 3))] = c0[int(rng.integers(0, 3))]                # sometimes flat in a channel
                field = ((xx * 5 + yy * 3 + (xx * yy) // 3) % 23) / 22.0                      # not bilinear
                if rng.integers(0, 2):
                    field = np.sqrt(((xx - 7.5) ** 2 + (yy - 7.5) ** 2) / 112.5)
                idx = np.clip(np.floor(field * (len(pts) - 1) + 0.5), 0, len(pts) - 1).astype(np.int64)
                flip = rng.integers(0, 2, 3)
                for c in range(3):
                    v = pts[idx, c]
                    if flip[c]:
                        v = 1.0 - v
                    out[c][sl[1:]] = np.floor(c0[c] + v * (c1[c] - c0[c]) + 0.5)
    return np.ascontiguousarray(np.clip(out, 0, 255).astype(np.int32))


def random_bank(seed: int, n_patterns: int = 5) -> list:
    """Banks no curve designer would draw: random point counts (1..64), unordered points, duplicates, clusters -- the tables and the search
    must agree with the reference on them too."""
    rng = np.random.default_rng(seed)
    pats = []
    for k in range(n_patterns):
        n = int(rng.integers(1, 65))
        if k % 3 == 0:
            p = rng.integers(0, 64, (n, 3))
        elif k % 3 == 1:
            t = np.sort(rng.random(n))
            p = np.stack([63 * t, 63 * t ** float(rng.uniform(0.4, 2.5)), 63 * (1 - t) ** float(rng.uniform(0.4, 2.5))], 1) + rng.integers(-2, 3, (n, 3))
        else:
            c = rng.integers(8, 56, (4, 3))
            p = c[rng.integers(0, 4, n)] + rng.integers(-6, 7, (n, 3))
        pats.append(np.clip(np.floor(p + 0.5), 0, 63).astype(np.uint8))
    return pats


def bank16() -> list:
    """Sixteen patterns: the six designed curves + ten of the random kind (point counts 1..64, unordered, duplicated, clustered)."""
    return bank_patterns() + random_bank(7, 10)


def lut_image_rgba(w: int, h: int, patterns=None, seed: int = 3) -> np.ndarray:
    """lut_image plus an alpha plane: transparent frame and holes on the 16x16 grid (the alpha tile-reject in front of the LUT search)."""
    rgb = lut_image(w, h, patterns, seed)
    y, x = np.mgrid[0:h, 0:w]
    hole = (x < 16) | (y >= h - 32) | ((((x >> 4) + 2 * (y >> 4)) % 7) == 0)
    return np.ascontiguousarray(np.concatenate([rgb, np.where(hole, 0, 255).astype(np.int32)[None]]))
