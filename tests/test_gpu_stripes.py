"""Row-stripe sharding on one GPU: every stripe is encoded by its own handle exactly as a rank would
(owned rows + 1 halo row, stripe bbox -> host min/max combine -> yk_alpha_finish), and the concatenation of the per-stripe
tile maps must equal the whole-image result bit for bit.  This is what makes the N > 1 path correct by construction."""
import numpy as np
import pytest

from tests.images import edge_image, synth_planes
from yaik_amd import distributed as ykd

pytestmark = pytest.mark.gpu


def _encode_whole(planes, m3):
    from yaik_amd.encoder import HipTileEncoder
    e = HipTileEncoder(0)
    e.set_image(planes)
    a = e.mip_prefilter() if planes.shape[0] == 4 else None
    e.encode(3, m3, False)
    out = {"alpha": a, "bitmaps": [e.gradient_bitmap(i) for i in range(7)], "cov": e.coverage(),
           "range": [e.range_streams(p) for p in range(3)], "corners": [e.gradient_corners(i) for i in range(7)]}
    e.close()
    return out


@pytest.mark.parametrize("case,world", [("synth512", 2), ("synth512", 4), ("synth512", 8), ("mixed256x384", 3), ("rgb512", 4)])
def test_stripes_concatenate_to_whole_image(case, world):
    from yaik_amd.encoder import HipTileEncoder
    planes = {"synth512": lambda: synth_planes(512, n_planes=4), "rgb512": lambda: synth_planes(512, n_planes=3),
              "mixed256x384": lambda: edge_image(256, 384, "mixed", 4)}[case]()
    n, H, W = planes.shape
    whole = _encode_whole(planes, False)
    encs, boxes = [], []
    for r in range(world):
        y0, h, halo = ykd.stripe_rows(H, world, r)
        if h == 0:
            encs.append(None); continue
        e = HipTileEncoder(0)
        e.set_image(np.ascontiguousarray(planes[:, y0:y0 + h + halo, :]), full_h=H, y0=y0, halo_rows=halo)
        if n == 4:
            e.alpha_reject()
            boxes.append(e.stripe_bbox())
        encs.append(e)
    gb = ykd.combine_bboxes(boxes) if n == 4 else None
    bitmaps = [[] for _ in range(7)]
    covs, defs, nibs, nns = [], [[], [], []], [[], [], []], [[], [], []]
    corner_streams, corner_edges = [], []
    abits, remaining = None, 0
    for e in encs:
        if e is None:
            continue
        if n == 4:
            e.alpha_finish(gb)
            ar = e.alpha_result()
            abits = ar["bitmap"] if abits is None else (abits | ar["bitmap"])
            remaining += ar["remaining"]
            assert np.array_equal(ar["bounds"], whole["alpha"]["bounds"])
        e.encode(3, False, False)
        for i in range(7):
            bitmaps[i].append(e.gradient_bitmap(i))
        covs.append(e.coverage())
        corner_streams.append([e.gradient_corners(i) for i in range(7)])
        corner_edges.append(e.gradient_corner_edges())
        for p in range(3):
            d, nb, nn = e.range_streams(p)
            defs[p].append(d); nibs[p].append(nb); nns[p].append(nn)
        e.close()
    for i in range(7):
        assert np.array_equal(np.concatenate(bitmaps[i]), whole["bitmaps"][i]), f"bitmap {i}"
    assert np.array_equal(np.concatenate(covs, axis=0), whole["cov"])
    for p in range(3):
        wd, wn, wnn = whole["range"][p]
        assert np.array_equal(np.concatenate(defs[p]), wd)
        cat, total = ykd.concat_nibble_streams(nibs[p], nns[p])
        assert total == wnn and np.array_equal(cat, wn)
    # corner-colour streams: stripe-local de-duplication + root-side reconciliation of the shared lattice rows
    merged = ykd.merge_corner_streams(corner_streams, corner_edges)
    for i in range(7):
        assert np.array_equal(merged[i], whole["corners"][i]), f"corner stream {i}: {merged[i].size} vs {whole['corners'][i].size}"
    if n == 4:
        assert np.array_equal(abits, whole["alpha"]["bitmap"])
        assert remaining == whole["alpha"]["remaining"]
