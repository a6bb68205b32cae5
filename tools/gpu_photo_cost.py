"""Fused-kernel cost on photo-like synthetic content (timing + the first 512x512 crop checked bit-exact against the oracle):
smooth illumination + a few hard-edged objects + band-limited texture of varying strength + sensor-like noise in part of the frame.
usage: gpu_photo_cost.py [W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yaik_amd.encoder import HipTileEncoder

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(7)


def photo(w):
    y = torch.arange(w, device=dev, dtype=torch.float32)[:, None] / w
    x = torch.arange(w, device=dev, dtype=torch.float32)[None, :] / w
    out = []
    for c in range(3):
        base = 110 + 70 * torch.sin(2.1 * x + 0.7 * c) * torch.cos(1.7 * y - 0.4 * c) + 40 * (x - y)      # smooth illumination
        tex = torch.zeros_like(base)
        for k in range(1, 6):                                                                              # band-limited texture
            fx, fy = 37.0 * k + 11 * c, 29.0 * k + 5 * c
            tex = tex + (1.0 / k) * torch.sin(2 * np.pi * (fx * x + fy * y) + k)
        strength = 14 * torch.clamp(torch.sin(3 * np.pi * x) * torch.sin(2 * np.pi * y), min=0) ** 2       # textured regions only
        img = base + strength * tex
        for (x0, y0, x1, y1, v) in ((0.1, 0.15, 0.3, 0.45, 60), (0.55, 0.2, 0.9, 0.35, -50), (0.35, 0.6, 0.7, 0.9, 35)):   # objects
            m = ((x > x0) & (x < x1) & (y > y0) & (y < y1)).float()
            img = img + m * (v + 8 * c)
        noise = torch.randn(w, w, device=dev, generator=g) * (2.5 * (y > 0.5).float())                    # noisy lower half
        out.append(torch.clamp(torch.round(img + noise), 0, 255).to(torch.int32))
    return torch.stack(out).contiguous()


planes = photo(W)
enc = HipTileEncoder(0)
enc.set_image(planes)
for m3 in (False, True):
    for _ in range(2):
        enc.encode(3, m3, False)
    tot = 0.0
    for _ in range(5):
        enc.encode(3, m3, False); tot += enc.kernel_ms()["encode"]
    cov = float(enc.coverage_fraction()) if hasattr(enc, "coverage_fraction") else float("nan")
    print(f"photo-like {W}x{W} RGB mode3={int(m3)}: fused kernel {tot / 5:.4f} ms = {W * W / (tot / 5) / 1e6:.1f} Gpix/s; "
          f"accepted tiles per pass {enc.gradient_counts().tolist()}", flush=True)
# parity on a crop
from oracle import pyoracle
from tests.parity import compare_encode
pyoracle.build()
crop = planes[:, :512, :512].cpu().numpy().copy()
e2 = HipTileEncoder(0)
bad = compare_encode(crop, e2, False, want_dst=False, check_corners=True)
print("parity of the 512x512 crop vs the oracle:", "bit-exact" if not bad else bad[:3])
