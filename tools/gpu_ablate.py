"""Timing-only ablation of the fused kernels on the GPU box: which phase costs what (results are wrong while flags != 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yaik_amd._lib import test_lib as lib           # ablation switches live in the test build of the library (include/yaik_hip_test.h)
from yaik_amd.encoder import HipTileEncoder
from yaik_amd.synth import synth_planes_torch

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
planes = synth_planes_torch(W, n_planes=4, device="cuda")
enc = HipTileEncoder(0, hooks=True)
enc.set_image(planes)
enc.alpha_reject(); enc.alpha_finish(None)
names = {0: "full", 1: "no range", 2: "no gradient", 3: "load+stage only", 4: "no table gathers"}
for ver in (2,):
    lib().yk_set_kernel_version(enc._h, ver)
    for flags, name in names.items():
        lib().yk_set_ablation(enc._h, flags)
        for _ in range(2):
            enc.encode(3, False, False)
        tot = 0.0
        for _ in range(5):
            enc.encode(3, False, False)
            tot += enc.kernel_ms()["encode"]
        print(f"kernel v{ver} ablate={flags:2d} {name:18s} {tot/5:.4f} ms", flush=True)
    lib().yk_set_ablation(enc._h, 0)
    tot = 0.0
    for _ in range(5):
        enc.encode(3, True, False)
        tot += enc.kernel_ms()["encode"]
    print(f"kernel v{ver} mode3BitOnly                    {tot/5:.4f} ms", flush=True)
