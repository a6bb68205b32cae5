"""First-light script for the GPU box: runs a few parity cases and prints mismatches verbosely."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle
pyoracle.build()
from tests.images import edge_image, synth_planes
from tests.parity import compare_encode
from yaik_amd.encoder import HipTileEncoder

hip = HipTileEncoder(0)
cases = [("synth64x3", synth_planes(64, n_planes=3)), ("noise64", edge_image(64, 64, "noise")), ("flat64", edge_image(64, 64, "flat")),
         ("synth256x4", synth_planes(256, n_planes=4)), ("mixed72x40", edge_image(72, 40, "mixed")), ("synth1024x4", synth_planes(1024, n_planes=4))]
allok = True
for name, pl in cases:
    for m3 in (False, True):
        t = time.time()
        bad = compare_encode(pl, hip, m3)
        print(name, "m3" if m3 else "m0", "OK" if not bad else "BAD", f"{time.time()-t:.2f}s", flush=True)
        for b in bad[:12]:
            print("    ", b)
        allok &= not bad
print("ALL OK" if allok else "FAILED")
