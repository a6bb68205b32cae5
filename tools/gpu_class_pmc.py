"""One content class of YAIK-synth v1 as a whole 8192x8192 RGB frame, encoded 3 times: run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES` to get the instruction budget per class.
usage: gpu_class_pmc.py ramp|mild|noise|mild16|mild32|mild64"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yaik_amd.encoder import HipTileEncoder
from yaik_amd._lib import lib

cls = sys.argv[1]
abl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W = 8192
dev = "cuda"
x = torch.arange(W, device=dev, dtype=torch.int64)[None, :].expand(W, W)
y = torch.arange(W, device=dev, dtype=torch.int64)[:, None].expand(W, W)
img = torch.stack([(255 * x) // W, (255 * y) // W, (255 * (x + y)) // (2 * W)])
g = torch.Generator(device=dev); g.manual_seed(1)
if cls.startswith("mild"):
    amp = int(cls[4:] or 8)
    img = (img + torch.randint(0, amp, (3, W, W), device=dev, generator=g)) % 256
elif cls == "noise":
    img = torch.randint(0, 256, (3, W, W), device=dev, generator=g)
if cls == "frame":
    from yaik_amd.synth import synth_planes_torch
    img = synth_planes_torch(W, W, 4, device=dev)
enc = HipTileEncoder(0, hooks=abl != 0)              # ablation switches live in the test build of the library
if abl:
    from yaik_amd._lib import test_lib
    test_lib().yk_set_ablation(enc._h, abl)
enc.set_image(img.to(torch.int32).contiguous())
if cls == 'frame':
    enc.alpha_reject(); enc.alpha_finish(None)
tot = 0.0
for i in range(4):
    enc.encode(3, False, False)
    if i: tot += enc.kernel_ms()["encode"]
print(f"{cls} ablate={abl}: fused kernel {tot / 3:.4f} ms", flush=True)
