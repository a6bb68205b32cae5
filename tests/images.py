"""Deterministic test images shared by the CPU and GPU parity tests (int32 planes [n, h, w], values 0..255)."""
import numpy as np

from yaik_amd.synth import synth_planes  # noqa: F401  (re-export)


def natural_photo() -> np.ndarray:
    """A 256x256 crop of a natural photograph (NASA portrait of Eileen Collins, public domain, as shipped in scikit-image's data/
    astronaut.png), committed as tests/golden/photo_astronaut256_input.npz by tests/golden/make_golden.py: the one test input that
    is neither synthetic nor procedural."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "photo_astronaut256_input.npz")
    rgb = np.load(path)["rgb"]
    return np.ascontiguousarray(rgb.transpose(2, 0, 1).astype(np.int32))


def lineart_image(size: int = 1024, n_planes: int = 4) -> np.ndarray:
    """Procedurally drawn flat-colour illustration (the content YAIK targets, README.md:1-13): cel-shaded shapes with anti-aliased
    edges, dark outlines, thin strokes, soft in-shape gradients over a two-tone background; with 4 planes the figure sits on a
    transparent surround with an anti-aliased alpha edge.  Only +, *, / and sqrt on float64 (bit-reproducible everywhere)."""
    n = size
    y, x = np.mgrid[0:n, 0:n].astype(np.float64)
    u, v = x / n, y / n

    def cover(d):                                   # signed distance (pixels, negative inside) -> anti-aliased coverage
        return np.clip(0.5 - d, 0.0, 1.0)

    def disc(cx, cy, r):
        return np.sqrt((x - cx * n) ** 2 + (y - cy * n) ** 2) - r * n

    def ellipse(cx, cy, rx, ry):
        q = np.sqrt(((x - cx * n) / (rx * n)) ** 2 + ((y - cy * n) / (ry * n)) ** 2)
        return (q - 1.0) * min(rx, ry) * n

    def box(cx, cy, hw, hh, rad):
        dx = np.abs(x - cx * n) - (hw - rad) * n
        dy = np.abs(y - cy * n) - (hh - rad) * n
        return np.sqrt(np.maximum(dx, 0) ** 2 + np.maximum(dy, 0) ** 2) + np.minimum(np.maximum(dx, dy), 0) - rad * n

    def segment(x0, y0, x1, y1, half):
        px, py, qx, qy = x0 * n, y0 * n, x1 * n, y1 * n
        t = np.clip(((x - px) * (qx - px) + (y - py) * (qy - py)) / ((qx - px) ** 2 + (qy - py) ** 2), 0.0, 1.0)
        return np.sqrt((x - (px + t * (qx - px))) ** 2 + (y - (py + t * (qy - py))) ** 2) - half

    img = np.empty((3, n, n))
    sky_top, sky_bot = np.array([236.0, 204.0, 168.0]), np.array([120.0, 176.0, 232.0])
    for c in range(3):
        img[c] = sky_top[c] + (sky_bot[c] - sky_top[c]) * v                      # smooth vertical gradient

    def paint(d, colour, outline=0.0, shade=None):
        nonlocal img
        a = cover(d)
        col = np.array(colour, dtype=np.float64)[:, None, None] * np.ones((1, n, n))
        if shade is not None:                                                    # cel shading: a second flat tone on one side
            sd, scol = shade
            sa = cover(sd)
            col = col * (1 - sa) + np.array(scol, dtype=np.float64)[:, None, None] * sa
        img = img * (1 - a) + col * a
        if outline > 0:
            o = cover(np.abs(d) - outline)
            img = img * (1 - o) + np.array([28.0, 20.0, 36.0])[:, None, None] * o

    # far hills (flat tones), a sun with a soft radial falloff, a figure made of a few shapes, strokes for hair / grass
    paint(ellipse(0.25, 1.02, 0.55, 0.30), (96, 148, 110))
    paint(ellipse(0.80, 1.05, 0.50, 0.26), (70, 122, 96))
    sun = disc(0.78, 0.20, 0.085)
    glow = np.clip(1.0 - np.sqrt((u - 0.78) ** 2 + (v - 0.20) ** 2) / 0.30, 0.0, 1.0) ** 2
    for c, g in enumerate((40.0, 30.0, 6.0)):
        img[c] = np.minimum(img[c] + g * glow, 255.0)
    paint(sun, (255, 244, 200))
    body = box(0.42, 0.70, 0.13, 0.20, 0.05)
    paint(body, (214, 72, 88), outline=1.6, shade=(segment(0.50, 0.50, 0.52, 0.92, 0.045 * n), (168, 48, 70)))
    head = disc(0.42, 0.40, 0.115)
    paint(head, (250, 222, 196), outline=1.6, shade=(disc(0.47, 0.44, 0.10) * -1.0 + 0.0 * x - 0.02 * n, (232, 190, 170)))
    paint(ellipse(0.42, 0.325, 0.135, 0.075), (64, 52, 110), outline=1.4)        # hair cap
    for k in range(9):                                                           # hair strands: thin anti-aliased strokes
        x0 = 0.30 + 0.03 * k
        paint(segment(x0, 0.33, x0 - 0.03 + 0.008 * k, 0.47 + 0.004 * k * k, 1.1 + 0.15 * k), (52, 40, 96))
    paint(disc(0.385, 0.405, 0.014), (40, 30, 60)); paint(disc(0.455, 0.405, 0.014), (40, 30, 60))     # eyes
    paint(segment(0.395, 0.455, 0.445, 0.457, 1.2), (150, 60, 70))                                      # mouth
    for k in range(24):                                                          # grass strokes of varying width
        gx = 0.04 + 0.04 * k
        paint(segment(gx, 0.99, gx + 0.012 - 0.001 * k, 0.90 - 0.002 * (k % 5) * k / 4, 0.8 + 0.1 * (k % 4)), (40, 96 + 3 * k, 60))
    rgb = np.clip(np.round(img), 0, 255)
    planes = [rgb[0], rgb[1], rgb[2]]
    if n_planes == 4:
        # opaque inside a rounded panel with an anti-aliased border, transparent surround (so the kept-tile box shrinks)
        panel = box(0.50, 0.53, 0.36, 0.40, 0.06)
        a = np.round(255.0 * cover(panel))
        a = np.where(disc(0.16, 0.80, 0.05) < 0, 0.0, a)                          # a hole
        planes.append(a)
    return np.ascontiguousarray(np.stack(planes).astype(np.int32))


def edge_image(w: int, h: int, kind: str, n_planes: int = 3, seed: int = 7) -> np.ndarray:
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    if kind == "flat":
        rgb = np.stack([np.full((h, w), 200), np.full((h, w), 17), np.full((h, w), 90)])
    elif kind == "noise":
        rgb = rng.integers(0, 256, (3, h, w))
    elif kind == "ramp":
        rgb = np.stack([(x * 2) % 256, (y * 3) % 256, (x + y) % 256])
    elif kind == "smooth":
        rgb = np.stack([(x * 255) // w, (y * 255) // h, ((x + y) * 255) // (w + h)]) + rng.integers(0, 3, (3, h, w))
    elif kind == "white":
        rgb = np.full((3, h, w), 255)
        rgb[:, h // 2:, :] = rng.integers(250, 256, (3, h - h // 2, w))
    elif kind == "dark":
        rgb = rng.integers(0, 3, (3, h, w))
        rgb[:, : h // 2, :] = 0
    elif kind == "mixed":
        rgb = np.stack([(x * 255) // w, (y * 255) // h, ((x + y) * 255) // (w + h)])
        m = ((x // 16 + y // 16) % 3) == 0
        rgb = np.where(m, rng.integers(0, 256, (3, h, w)), rgb)
        m2 = ((x // 8 + y // 8) % 5) == 0
        rgb = np.where(m2, np.clip(rgb + rng.integers(-6, 7, (3, h, w)), 0, 255), rgb)
    elif kind == "photo":
        # photo-like: smooth illumination + hard-edged objects + band-limited texture of varying strength + mild sensor noise:
        # 8x8 tiles with every span from 0 to ~150, i.e. every rangeDecode regime of DynamicTile::buildTable
        fx, fy = x / w, y / h
        chans = []
        for c in range(3):
            base = 110 + 70 * np.sin(2.1 * fx + 0.7 * c) * np.cos(1.7 * fy - 0.4 * c) + 40 * (fx - fy)
            tex = np.zeros_like(base)
            for k in range(1, 6):
                tex += (1.0 / k) * np.sin(2 * np.pi * ((5.0 * k + 2 * c) * fx * (w / 64) + (4.0 * k + c) * fy * (h / 64)) + k)
            strength = 40 * np.clip(np.sin(3 * np.pi * fx) * np.sin(2 * np.pi * fy), 0, None) ** 2
            img = base + strength * tex
            for (x0, y0, x1, y1, v) in ((0.1, 0.15, 0.3, 0.45, 60), (0.55, 0.2, 0.9, 0.35, -50), (0.35, 0.6, 0.7, 0.9, 35)):
                img = img + ((fx > x0) & (fx < x1) & (fy > y0) & (fy < y1)) * (v + 8 * c)
            img = img + rng.normal(0, 2.5, (h, w)) * (fy > 0.5)
            chans.append(np.round(img))
        rgb = np.stack(chans)
    elif kind == "planemix":
        # per 16x16 block a subset of the channels is a smooth ramp and the rest noise: what the partial-plane gradient passes
        # (FittingQuadSmooth with NULL planes) exist for; every subset 0..7 occurs, with block-aligned and 4-pixel-shifted borders
        ramp = np.stack([(x * 255) // w, (y * 255) // h, ((x + y) * 255) // (w + h)])
        sel = (((x + 4 * ((y // 32) & 1)) // 16) * 3 + (y // 16) * 5) % 8
        noise = rng.integers(0, 256, (3, h, w))
        rgb = np.stack([np.where((sel >> c) & 1, ramp[c], noise[c]) for c in range(3)])
    elif kind == "twocolor":
        a = rng.integers(0, 256, 3); b = rng.integers(0, 256, 3)
        sel = ((x * 7 + y * 13) // 5) % 2
        rgb = np.stack([np.where(sel, a[c], b[c]) for c in range(3)])
    else:
        raise KeyError(kind)
    rgb = np.clip(rgb, 0, 255)
    planes = [rgb[0], rgb[1], rgb[2]]
    if n_planes == 4:
        a = np.full((h, w), 255)
        a[:, : max(16, w // 8)] = 0
        a[: max(16, h // 16), :] = 0
        a[(x // 16 % 4 == 1) & (y // 16 % 3 == 1)] = 0
        if w >= 64 and h >= 64:
            a[40:56, 40:57] = np.where(rng.integers(0, 4, (16, 17)) == 0, 7, 0)
        planes.append(a)
    return np.ascontiguousarray(np.stack(planes).astype(np.int32))
