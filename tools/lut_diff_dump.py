"""dump the 3-D LUT search results of a test case for the library named by YK_LIB (debugging aid): python tools/lut_diff_dump.py out.npz"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_lut3d import CASES
from tests.blobs import LUT_PASSES
from yaik_amd.encoder import HipTileEncoder

planes, pats = CASES["full_bank_64"]()
hip = HipTileEncoder(0)
hip.lut_clear()
for p in pats:
    hip.lut_load(p)
hip.set_image(planes)
hip.encode(3, False, False)
hip.lut_start()
counts = [hip.lut_search(sx, sy) for sx, sy in LUT_PASSES]
s = hip.lut_streams()
np.savez(sys.argv[1], counts=np.array(counts), **s)
print(counts)
