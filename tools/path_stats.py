"""How often every path of yk_encode2_kernel is taken on one frame (needs a -DYK2_STATS build: YK_LIB=build/libS.so).
usage: YK_LIB=build/libS.so python tools/path_stats.py [frame|ramp|mild|noise]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yaik_amd.encoder import HipTileEncoder
from yaik_amd._lib import lib

cls = sys.argv[1] if len(sys.argv) > 1 else "frame"
W = 8192
dev = "cuda"
if cls == "frame":
    from yaik_amd.synth import synth_planes_torch
    img = synth_planes_torch(W, W, 4, device=dev)
else:
    x = torch.arange(W, device=dev, dtype=torch.int64)[None, :].expand(W, W)
    y = torch.arange(W, device=dev, dtype=torch.int64)[:, None].expand(W, W)
    img = torch.stack([(255 * x) // W, (255 * y) // W, (255 * (x + y)) // (2 * W)])
    g = torch.Generator(device=dev); g.manual_seed(1)
    if cls == "mild": img = (img + torch.randint(0, 8, (3, W, W), device=dev, generator=g)) % 256
    elif cls == "noise": img = torch.randint(0, 256, (3, W, W), device=dev, generator=g)
enc = HipTileEncoder(0)
enc.set_image(img.to(torch.int32).contiguous())
if cls == "frame":
    enc.alpha_reject(); enc.alpha_finish(None)
L = lib()
L.yk_debug_path_stats.argtypes = [C.c_void_p, C.c_int]; L.yk_debug_path_stats.restype = C.c_int
st = np.zeros(128, np.uint64)
enc.encode(3, False, False); torch.cuda.synchronize()
assert L.yk_debug_path_stats(st.ctypes.data, 1) == 0
enc.encode(3, False, False); torch.cuda.synchronize()
assert L.yk_debug_path_stats(st.ctypes.data, 1) == 0
n = float(st[75])
names = ["16x16", "16x8", "8x16", "8x8", "8x4", "4x8", "4x4"]
print(f"{cls}: strips {int(n)}, all-dead strips {st[76] / n:.3f}, dead lanes per strip {st[77] / n:.1f}")
print("pass    entered compact  exit1   exit2   exit4x4 fullA   walkP   accept  cells/entered  compactCells restCells/walkP")
for p in range(7):
    r = st[p * 10:p * 10 + 10].astype(float)
    e = max(r[0], 1.0)
    print(f"{names[p]:6s} {r[0] / n:7.3f} {r[1] / n:7.3f} {r[2] / n:7.3f} {r[3] / n:7.3f} {r[9] / n:7.3f} {r[4] / n:7.3f} {r[5] / n:7.3f} {r[6] / n:7.3f} "
          f"{r[8] / e:8.1f} {r[7] / max(r[1], 1):8.1f} {float(st[90 + p]) / max(r[5], 1):8.1f}")
print("compact: exits after stage 1 per strip", [round(float(st[100 + p]) / n, 3) for p in range(7)], " accepts", [round(float(st[110 + p]) / n, 3) for p in range(7)])
print("smooth 16x16 strips: tiles viable / accepted by raw|Round6 / by Round6P / by either, per strip:", [round(float(st[i]) / n, 3) for i in (107, 97, 98, 99)])
print("valid lanes per strip (0, 1-4, 5-8, 9-16, 17-32, 33-63, 64):", [round(float(st[120 + i]) / n, 3) for i in range(7)])
print(f"range: strips coded {st[70] / n:.3f}, valid lanes per coded strip {st[74] / max(float(st[70]), 1):.1f}, plane-waves with ambiguous tiles {st[71] / n:.4f} "
      f"(tiles {st[72] / n:.4f} per strip), tie blocks {st[73] / n:.4f} per strip")
