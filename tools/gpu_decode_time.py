"""Decode-side kernel times at full size (run under rocprofv3 --kernel-trace --stats): 8192x8192 RGB, GPU encode -> GPU decode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yaik_amd.encoder import HipTileEncoder
from yaik_amd.decoder import HipTileDecoder
from yaik_amd.synth import synth_planes_torch

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
PASSES = [(4, 4), (4, 3), (3, 4), (3, 3), (3, 2), (2, 3), (2, 2)]
planes = synth_planes_torch(W, n_planes=3, device="cuda")
enc = HipTileEncoder(0); enc.set_image(planes); enc.encode(3, False, False)
bms = [enc.gradient_bitmap(i) for i in range(7)]
inv = (255 << 16) // 250
rgbs = [((enc.gradient_corners(i).astype(np.int64) * inv) >> 16).astype(np.uint8) for i in range(7)]
counts = enc.gradient_counts()
pix, typ = enc.dynamic_tile_compressor()
dec = HipTileDecoder(0)
for rep in range(3):
    t0 = time.perf_counter()
    dec.begin(W, W)
    for i, (sx, sy) in enumerate(PASSES):
        if counts[i]:
            dec.decompress_gradient(sx, sy, bms[i], rgbs[i])
    t1 = time.perf_counter()
    dec.decompress_1d(typ, pix)
    t2 = time.perf_counter()
    img = dec.image()
    t3 = time.perf_counter()
    print(f"rep {rep}: gradient x7 {1e3*(t1-t0):.2f} ms, 1-D {1e3*(t2-t1):.2f} ms (pix stream {pix.size/1e6:.0f} MB from host), image out {1e3*(t3-t2):.2f} ms", flush=True)
