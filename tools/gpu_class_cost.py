"""Cost of the fused kernel per content class of YAIK-synth v1 (timing only): whole 8192x8192 RGB frames made of ONE class."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yaik_amd.encoder import HipTileEncoder

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda"
x = torch.arange(W, device=dev, dtype=torch.int64)[None, :].expand(W, W)
y = torch.arange(W, device=dev, dtype=torch.int64)[:, None].expand(W, W)
ramp = torch.stack([(255 * x) // W, (255 * y) // W, (255 * (x + y)) // (2 * W)])
g = torch.Generator(device=dev); g.manual_seed(1)
mild = (ramp + torch.randint(0, 8, (3, W, W), device=dev, generator=g)) % 256
noise = torch.randint(0, 256, (3, W, W), device=dev, generator=g)
enc = HipTileEncoder(0)
for name, img in (("ramp", ramp), ("mild noise", mild), ("noise", noise)):
    planes = img.to(torch.int32).contiguous()
    enc.set_image(planes)
    for m3 in (False, True):
        for _ in range(2):
            enc.encode(3, m3, False)
        tot = 0.0
        for _ in range(5):
            enc.encode(3, m3, False); tot += enc.kernel_ms()["encode"]
        print(f"{name:12s} mode3={int(m3)}  {tot / 5:.4f} ms   accepted tiles per pass {enc.gradient_counts().tolist()}", flush=True)
