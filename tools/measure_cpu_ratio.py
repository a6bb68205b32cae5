"""BASELINE.md §3 step 1 (run in the build container, where /root/reference exists): time the compiled, unmodified reference
and the repo's CPU restatement (oracle) on the same YAIK-synth v1 frames, same stages (MipPrefilter + 7x FittingQuadSmooth +
3x DynamicTileEncode 4-bpp), one thread each, and record both rates and their ratio in profiles/cpu_ratio.json.  The reference's
stages include what it does inside them (ZStd 18/21, PaletteCompressor, per-tile printf, debug PNGs); the restatement does not,
which is most of the ratio.  bench.py quotes the ratio next to the CPU figure it measures on the GPU box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from oracle.refrun import have_ref, run_reference  # noqa: E402
from yaik_amd.synth import synth_planes  # noqa: E402


def main():
    if not have_ref():
        print("oracle/_ref/ref_driver missing", file=sys.stderr)
        return 1
    pyoracle.build()
    os.environ["YK_REF_KEEP_LEVEL"] = "1"
    rows = []
    for size in (1024, 2048):
        planes = synth_planes(size, n_planes=4)
        ref_t = []
        for _ in range(3):
            st = np.frombuffer(run_reference(planes)["stage_seconds"], np.float64)
            ref_t.append(float(st.sum()))
        ora_t = []
        for _ in range(3):
            o = pyoracle.OracleEncoder(planes)
            t0 = time.perf_counter()
            o.mip_prefilter()
            for sx, sy in pyoracle.PASSES:
                o.fitting_quad_smooth(sx, sy)
            for p in range(3):
                o.dynamic_tile_encode(p, False)
            ora_t.append(time.perf_counter() - t0)
        mpix = size * size / 1e6
        r, q = sorted(ref_t)[1], sorted(ora_t)[1]
        rows.append({"size": size, "reference_s": round(r, 3), "reference_mpix_s": round(mpix / r, 3), "port_s": round(q, 3),
                     "port_mpix_s": round(mpix / q, 3), "port_over_reference": round(r / q, 2)})
        print(rows[-1])
    out = {"what": "median of 3, one thread, build container (8 vCPU Xeon, shared), YAIK-synth v1 RGBA, stages: MipPrefilter + 7x FittingQuadSmooth + "
                   "3x DynamicTileEncode (4-bpp); reference = unmodified sources incl. their ZStd 18/21, PaletteCompressor, printf and debug PNG work",
           "rows": rows, "port_over_reference": rows[-1]["port_over_reference"]}
    with open(os.path.join(ROOT, "profiles", "cpu_ratio.json"), "w") as f:
        json.dump(out, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
