"""CPU suite: the oracle restatement against the committed golden vectors captured from the unmodified reference
(tests/golden/*.npz + hashes.json, generator tests/golden/make_golden.py).  Runs without /root/reference and without a GPU."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests.blobs import SKIP, compare_with_reference, oracle_blobs
from tests.golden.make_golden import FULL, HASHED

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.mark.parametrize("name", sorted(FULL))
def test_oracle_matches_full_fixture(oracle_built, name):
    ref = dict(np.load(os.path.join(GOLD, name + ".npz")))
    ref = {k: v.tobytes() for k, v in ref.items()}
    planes = FULL[name]()
    n, h, w = planes.shape
    bad = compare_with_reference(ref, oracle_blobs(planes), decode_ok=(w % 16 == 0 and h % 16 == 0))
    assert not bad, bad


@pytest.mark.parametrize("name", sorted(HASHED))
def test_oracle_matches_hashed_fixture(oracle_built, name):
    with open(os.path.join(GOLD, "hashes.json")) as f:
        want = json.load(f)[name]
    planes = HASHED[name]()
    ours = oracle_blobs(planes)
    bad = []
    for k, hsh in want.items():
        if k in SKIP or k == "grad_counts_values" or k not in ours:
            continue
        if hashlib.sha256(ours[k]).hexdigest() != hsh:
            bad.append(k)
    assert not bad, bad
    assert np.frombuffer(ours["grad_counts"], np.int32).tolist() == want["grad_counts_values"]
