"""Per-wave phase timeline of yk_encode2_kernel on the bench frame (needs the -DYK2_TIMING build of the library, see
tools/wave_timeline.sh).  Every wave records the shader clock and the 100 MHz real-time counter at: entry, pixels staged,
gradient passes done, range phase entered, exit.  Prints, per content class of YAIK-synth v1, the mean time per phase, and the
number of waves in flight over the life of the launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yaik_amd._lib import lib
from yaik_amd.encoder import HipTileEncoder
from yaik_amd.synth import synth_planes_torch

W = 8192
planes = synth_planes_torch(W, n_planes=4, device="cuda")
enc = HipTileEncoder(0)
enc.set_image(planes)
enc.alpha_reject(); enc.alpha_finish(None)
for _ in range(3):
    enc.encode(3, False, False)
torch.cuda.synchronize()
L = lib()
L.yk_debug_wave_times.argtypes = [C.c_void_p]; L.yk_debug_wave_times.restype = C.c_int
t = np.zeros((65536, 8, 2), np.uint64)
assert L.yk_debug_wave_times(t.ctypes.data) == 0
clk = t[:, :5, 0].astype(np.int64); rt = t[:, :5, 1].astype(np.int64)
rt0 = rt[:, 0].min()
us = (rt - rt0) / 100.0                                  # 100 MHz counter -> microseconds since the first wave started
print(f"launch span {us[:, 4].max():.1f} us; shader clock while running: {np.median((clk[:, 4] - clk[:, 0]) / np.maximum(rt[:, 4] - rt[:, 0], 1)) * 100:.0f} MHz")
unit = np.arange(65536); blk = unit >> 2; BY, BX = blk // 128, blk % 128
k = (BX + BY) & 3
x0, y0 = BX * 64, BY * 64
kept = ~((x0 < W // 8) | (x0 >= W - W // 16) | (y0 < W // 16) | (y0 >= W - W // 8) | ((((x0 >> 7) + (y0 >> 7)) % 5) == 0))
names = {0: "ramp (right neighbour ramp)", 1: "ramp next to mild noise", 2: "mild noise", 3: "noise"}
print(f"{'class':38s} {'strips':>7s} {'load':>7s} {'gradient':>9s} {'outputs':>8s} {'range':>7s} {'total':>7s}  (mean us per wave)")
tot_wave_us = 0.0
for kk in range(4):
    for kp in (True, False):
        m = (k == kk) & (kept == kp)
        d = np.diff(us[m], axis=1).mean(axis=0)
        tot_wave_us += np.diff(us[m], axis=1).sum()
        print(f"{names[kk] + (', kept' if kp else ', alpha-rejected'):38s} {int(m.sum()):7d} {d[0]:7.2f} {d[1]:9.2f} {d[2]:8.2f} {d[3]:7.2f} {d.sum():7.2f}")
print(f"sum of wave lifetimes {tot_wave_us / 1e3:.1f} ms = {tot_wave_us / us[:, 4].max() :.0f} waves in flight on average (4096 slots)")
edges = np.linspace(0, us[:, 4].max(), 21)
act = [(int(((us[:, 0] <= e) & (us[:, 4] > e)).sum())) for e in edges]
print("waves in flight at 5 % steps of the launch:", act)
life = us[:, 4] - us[:, 0]
order = np.argsort(-life)[:12]
print("longest waves: (unit, class, kept, start us, load, gradient, outputs, range)")
for u in order:
    d = np.diff(us[u])
    print(f"  {u:6d} k={k[u]} kept={int(kept[u])} start {us[u,0]:7.1f}  {d[0]:6.1f} {d[1]:6.1f} {d[2]:6.1f} {d[3]:6.1f}")
for q in (50, 90, 99, 99.9):
    print(f"lifetime p{q}: {np.percentile(life, q):.1f} us", end="; ")
print()
late = us[:, 4] > 0.86 * us[:, 4].max()
print("waves still running after 86 % of the launch:", int(late.sum()), "; their classes:", np.bincount(k[late] * 2 + kept[late].astype(int), minlength=8).tolist(),
      "; start times (us) min/median:", float(us[late, 0].min()), float(np.median(us[late, 0])))
print("start time of the last wave:", float(us[:, 0].max()), "us; waves started per 5 % step:", np.histogram(us[:, 0], bins=edges)[0].tolist())
amb_tiles = t[:, 5, 0].astype(np.int64); amb_planes = t[:, 5, 1].astype(np.int64)
print("ambiguous tile-planes per wave (exact re-summation): waves with any:", int((amb_tiles > 0).sum()), "; histogram 0,1,2,3-4,5-8,9-16,17+:",
      [int(((amb_tiles >= a) & (amb_tiles <= b)).sum()) for a, b in ((0, 0), (1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 999))],
      "; planes with any per wave:", np.bincount(amb_planes, minlength=4).tolist())
for a, b in ((0, 0), (1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 999)):
    m = (amb_tiles >= a) & (amb_tiles <= b) & (k == 2) & kept
    if m.any(): print(f"  mild kept waves with {a}-{b} ambiguous tile-planes: {int(m.sum())}, mean range phase {np.diff(us[m], axis=1)[:, 3].mean():.1f} us")
entry = (t[:, 6, 0].astype(np.int64) - rt0) / 100.0
hw = t[:, 6, 1]
slot = hw & np.uint64(0xFFFFFFFFFF)        # xcc | hw_id (wave, simd, cu, sh, se ...) identifies the hardware wave slot
print(f"entry -> first probe (kernel arguments, unit arithmetic): mean {np.mean(us[:, 0] - entry):.2f} us")
gaps = []
for sid in np.unique(slot):
    m = np.where(slot == sid)[0]
    o = m[np.argsort(entry[m])]
    g = entry[o][1:] - us[o, 4][:-1]
    gaps.append(g)
gaps = np.concatenate(gaps)
print(f"hardware wave slots seen: {len(np.unique(slot))}; idle gap between a wave's last probe and the next wave's entry on the same slot: "
      f"mean {gaps.mean():.2f} us, median {np.median(gaps):.2f}, p90 {np.percentile(gaps, 90):.2f}")
