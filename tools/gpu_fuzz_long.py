"""One-off long fuzz (not part of the suite): many more seeds of tests/test_gpu_fuzz_parity.py's tile-regime images, v2 kernel only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_fuzz_parity import _image
from tests.parity import compare_encode
from yaik_amd.encoder import HipTileEncoder
from oracle import pyoracle
pyoracle.build()
e = HipTileEncoder(0)
n0, n1 = int(sys.argv[1]), int(sys.argv[2])
bad_total = 0
for seed in range(n0, n1):
    size = (64, 128, 256)[seed % 3]
    planes = _image(5000 + seed, size, 4 if seed % 2 else 3)
    for m3 in (False, True):
        bad = compare_encode(planes, e, m3, want_dst=False, check_corners=True)
        if bad:
            bad_total += 1
            print("MISMATCH seed", seed, "m3", m3, bad[:3], flush=True)
print("seeds", n0, n1, "mismatching cases:", bad_total)
