"""One-off fuzz of the exact re-summation path (not part of the suite): ablation flag 16 sends EVERY tile through it, every fifth
seed also checks the reconstructed planes (wantDst, LUTs built the reference's way)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz_parity import _image
from tests.parity import compare_encode
from yaik_amd.encoder import HipTileEncoder
from yaik_amd._lib import lib
from oracle import pyoracle
pyoracle.build()
e = HipTileEncoder(0)
lib().yk_set_ablation(e._h, 16)
bad_total = 0
for seed in range(0, 300):
    size = (64, 128, 256)[seed % 3]
    planes = _image(9000 + seed, size, 4 if seed % 2 else 3)
    for m3 in (False, True):
        bad = compare_encode(planes, e, m3, want_dst=(seed % 5 == 0), check_corners=False)
        if bad:
            bad_total += 1; print("MISMATCH", seed, m3, bad[:2], flush=True)
print("exact-path fuzz, 300 seeds: mismatching cases:", bad_total)
