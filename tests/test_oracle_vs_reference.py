"""CPU suite: live comparison of the oracle restatement with the unmodified reference (oracle/_ref/ref_driver).
Skipped where the reference build is absent (e.g. a checkout without /root/reference); the golden-vector test covers that case."""
import pytest

from oracle.refrun import have_ref, run_reference
from tests.blobs import compare_with_reference, oracle_blobs
from tests.images import edge_image, lineart_image, natural_photo, synth_planes

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref/ref_driver not built (needs /root/reference)")

CASES = {
    "synth128_rgb": lambda: synth_planes(128, n_planes=3),
    "synth256_rgba": lambda: synth_planes(256, n_planes=4),
    "noise64_rgb": lambda: edge_image(64, 64, "noise", 3),
    "flat128_rgba": lambda: edge_image(128, 128, "flat", 4),
    "smooth128_rgba": lambda: edge_image(128, 128, "smooth", 4),
    "white128_rgb": lambda: edge_image(128, 128, "white", 3),
    "dark64_rgb": lambda: edge_image(64, 64, "dark", 3),
    "twocolor256_rgba": lambda: edge_image(256, 256, "twocolor", 4),
    "mixed200x136_rgb": lambda: edge_image(200, 136, "mixed", 3),
    "photo256_rgba": lambda: edge_image(256, 256, "photo", 4),
    "photo192x128_rgb": lambda: edge_image(192, 128, "photo", 3, seed=11),
    "lineart512_rgba": lambda: lineart_image(512, 4),            # drawn illustration, anti-aliased edges, alpha surround
    "photo_astronaut256_rgb": natural_photo,                     # natural photograph
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_equals_reference(oracle_built, name):
    planes = CASES[name]()
    n, h, w = planes.shape
    bad = compare_with_reference(run_reference(planes), oracle_blobs(planes), decode_ok=(w % 16 == 0 and h % 16 == 0))
    assert not bad, bad


@pytest.mark.parametrize("seed", [1000, 1003, 1004, 1007, 1009, 1014])
def test_oracle_equals_reference_on_fuzz_regimes(oracle_built, seed):
    """The tile regimes of tests/test_gpu_fuzz_parity.py (ranges around 16, values >= 223, zeros, two-level tiles, accept-boundary
    gradients): the oracle must agree with the unmodified reference there too, so the GPU fuzz test is anchored."""
    from tests.test_gpu_fuzz_parity import _image
    size = 256 if seed % 2 else 128                     # the reference's RGBA path needs >= 256 (bad_alloc below)
    planes = _image(seed, size, 4 if seed % 2 else 3)
    bad = compare_with_reference(run_reference(planes), oracle_blobs(planes))
    assert not bad, bad
