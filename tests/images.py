"""Deterministic test images shared by the CPU and GPU parity tests (int32 planes [n, h, w], values 0..255)."""
import numpy as np

from yaik_amd.synth import synth_planes  # noqa: F401  (re-export)


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
