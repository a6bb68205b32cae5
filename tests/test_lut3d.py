"""CPU suite, SURVEY 8(f)4: the 3-D LUT tile search (Load3DPattern / Set3DPointCloud / Correlation3DSearch / computeValues3D and the stream
finishing of EndCorrelationSearch), oracle vs the unmodified reference on a SYNTHETIC bank (tests/lutbank.py; the reference's own bank is
not in its repository) -- committed fixtures tests/golden/lut_*.npz, live where the reference build exists."""
import os

import numpy as np
import pytest

from oracle.refrun import have_ref, run_reference
from tests.blobs import compare_lut, oracle_lut_blobs
from tests.golden.make_golden import LUT3D
from tests.lutbank import bank_bytes, bank_patterns, lut_image, random_bank

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(LUT3D))
def test_oracle_lut_search_matches_fixture(oracle_built, name):
    ref = {k: v.tobytes() for k, v in np.load(os.path.join(GOLD, name + ".npz")).items()}
    planes, pats = LUT3D[name]()
    assert np.frombuffer(ref["lut_counts"], np.int32).reshape(6, 6)[-1, 0] > 0
    bad = compare_lut(ref, oracle_lut_blobs(planes, pats, tables=False), tables=False)
    assert not bad, bad


LIVE = {
    "lut128": lambda: (lut_image(128, 128, seed=11), bank_patterns()),
    "lut200x136": lambda: (lut_image(200, 136, seed=5), bank_patterns()),          # partial tiles on both edges
    "lut256_3patterns": lambda: (lut_image(256, 256, bank_patterns(3), seed=2), bank_patterns(3)),
    "random_bank_a": lambda: (lut_image(128, 128, random_bank(101), seed=41), random_bank(101)),
    "random_bank_b": lambda: (lut_image(144, 112, random_bank(102, 7), seed=42), random_bank(102, 7)),
    "random_bank_on_photo": lambda: (__import__("tests.images", fromlist=["edge_image"]).edge_image(128, 128, "photo", 3), random_bank(103, 4)),
}


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/ref_driver not built (needs /root/reference)")
@pytest.mark.parametrize("name", sorted(LIVE))
def test_oracle_lut_search_matches_reference_live(oracle_built, name):
    planes, pats = LIVE[name]()
    ours = oracle_lut_blobs(planes, pats)
    ref = run_reference(planes, lut_bank=bank_bytes(pats))
    if "lut_dec_planes" not in ours:                                          # the decode loops need whole 16x16 tiles
        ref = {k: v for k, v in ref.items() if not k.startswith(("lut_dec_", "dec_"))}
    bad = compare_lut(ref, ours)                                              # incl. every table of every pattern
    assert not bad, bad
