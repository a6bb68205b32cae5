"""CPU suite: the product's entropy stage and chunk framing (yaik_amd/host/chunks.cpp, palette.cpp, zstd_dl.cpp) against the
reference's own chunk stream — live where oracle/_ref is built, and from the committed golden fixtures everywhere."""
import glob
import os

import numpy as np
import pytest

from oracle.refrun import have_ref, run_reference
from tests import chunks
from tests.golden.make_golden import FULL
from tests.images import edge_image, synth_planes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tool(oracle_built):
    return chunks.build_tool()


@pytest.mark.parametrize("name", sorted(FULL))
def test_framed_stream_matches_golden_reference_stream(tool, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    if "chunks_file" not in z.files:
        pytest.skip("fixture predates chunks_file")
    planes = FULL[name]()
    n, h, w = planes.shape
    ref = chunks.parse(z["chunks_file"].tobytes(), w, h)
    ours = chunks.parse(chunks.frame(chunks.oracle_streams(planes)), w, h)
    bad = chunks.compare_parsed(ref, ours)
    assert not bad, bad


LIVE = {
    "synth128_rgb": lambda: synth_planes(128, n_planes=3),
    "smooth128_rgba": lambda: edge_image(128, 128, "smooth", 4),
    "twocolor256_rgba": lambda: edge_image(256, 256, "twocolor", 4),
    "mixed200x136_rgb": lambda: edge_image(200, 136, "mixed", 3),
    "synth512_rgba": lambda: synth_planes(512, n_planes=4),
}


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/ref_driver not built (needs /root/reference)")
@pytest.mark.parametrize("name", sorted(LIVE))
def test_framed_stream_matches_live_reference_stream(tool, name):
    planes = LIVE[name]()
    n, h, w = planes.shape
    ref = chunks.parse(run_reference(planes)["chunks_file"], w, h)
    ours = chunks.parse(chunks.frame(chunks.oracle_streams(planes)), w, h)
    bad = chunks.compare_parsed(ref, ours)
    assert not bad, bad


def test_file_header_and_terminator(tool):
    planes = synth_planes(64, n_planes=3)
    stream = chunks.frame(chunks.oracle_streams(planes), with_file_header=True)
    assert stream[:4] == b"YAIK" and stream[-4:] == bytes([0xEF, 0xBE, 0xAD, 0xDE]) and len(stream) % 4 == 0
    p = chunks.parse(stream, 0, 0)
    assert np.frombuffer(p["file_header"], np.int32).tolist() == [1, 64, 64, 0]
    assert np.frombuffer(p["chunk_count_terminated"], np.int32)[1] == 1
