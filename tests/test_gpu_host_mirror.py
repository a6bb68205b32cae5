"""The C++ drop-in of the reference's operator surface (yaik_amd/host: Plane / Image / EncoderContext with the same
method names and call order) driven by a small C++ program, compared blob by blob with the CPU oracle."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from oracle.refrun import parse_blobs
from tests.blobs import oracle_blobs
from tests.images import edge_image, synth_planes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "yaik_amd", "host", "host_driver")


def _run(planes, mode3):
    if not os.path.exists(DRIVER):
        subprocess.run(["make", "-C", os.path.dirname(DRIVER)], check=True)
    n, h, w = planes.shape
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.blobs")
        with open(fin, "wb") as f:
            f.write(struct.pack("<3i", w, h, n)); f.write(np.ascontiguousarray(planes, np.int32).tobytes())
        subprocess.run([DRIVER, fin, fout, "1" if mode3 else "0"], check=True)
        return parse_blobs(fout)


@pytest.mark.parametrize("case,mode3", [("synth256x4", False), ("synth256x4", True), ("mixed128x4", False), ("ramp72x40", False), ("synth512x3", False)])
def test_cpp_operator_surface_matches_oracle(oracle_built, case, mode3):
    planes = {"synth256x4": lambda: synth_planes(256, n_planes=4), "mixed128x4": lambda: edge_image(128, 128, "mixed", 4),
              "ramp72x40": lambda: edge_image(72, 40, "ramp", 3), "synth512x3": lambda: synth_planes(512, n_planes=3)}[case]()
    got = _run(planes, mode3)
    want = oracle_blobs(planes)
    from oracle.pyoracle import PASSES, OracleEncoder
    ora = OracleEncoder(planes)
    if planes.shape[0] == 4:
        ora.mip_prefilter()
    bad = []
    for i, (sx, sy) in enumerate(PASSES):
        cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy)
        if got[f"grad_bitmap_{i}"] != bm.tobytes():
            bad.append(f"bitmap{i}")
        if got[f"grad_rgbraw_{i}"] != rgb.tobytes():
            bad.append(f"rgb{i}")
    m = 1 if mode3 else 0
    names = ["grad_counts", "d1_pix", "d1_type"] + [f"plnt_{k}_{m}_{p}" for k in ("defs", "idx", "dst") for p in range(3)]
    if planes.shape[0] == 4:
        names += ["mip_bounds", "_mip_bitmap"]
    for k in names:
        if got[k] != want[k]:
            bad.append(k)
    assert not bad, bad
