"""The rate of the path when the boundary hands over HOST planes (yk_upload_planes: pageable numpy memory -> HBM over PCIe) and takes the packed
tile maps back to the host: never the bench's `value`, quoted in DESIGN.md next to it.  usage: python tools/pcie_inclusive.py [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yaik_amd.encoder import HipTileEncoder
from yaik_amd.synth import synth_planes_torch

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
host = synth_planes_torch(W, n_planes=4, device="cuda").cpu().numpy()
pinned = torch.from_numpy(host).pin_memory().numpy()
enc = HipTileEncoder(0)
enc.set_image(host)
blob = torch.empty(enc.export_capacity(), dtype=torch.uint8, device="cuda")
for name, planes in (("pageable host planes", host), ("pinned host planes", pinned)):
    for rep in range(3):
        t0 = time.perf_counter()
        enc.set_image(planes)                      # H2D upload of 16 B per pixel
        enc.alpha_reject(); enc.alpha_finish(None)
        enc.encode(3, False, False)
        sizes = enc.export_tile_maps(blob)         # packed tile maps (alpha bitmap, 7 tile bitmaps, 3 x defs, 3 x nibbles) in one device buffer
        n = int(sizes.sum())
        back = blob[:n].cpu()                      # ... and back on the host
        t1 = time.perf_counter()
    print(f"{name}: {1e3 * (t1 - t0):.2f} ms per {W}x{W} RGBA frame = {W * W / 1e6 / (t1 - t0):.0f} Mpix/s (upload {host.nbytes / 1e6:.0f} MB, maps back {n / 1e6:.1f} MB)", flush=True)
