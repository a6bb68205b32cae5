"""Kernel overlap of a bench command from a rocprofv3 --kernel-trace run: per kernel name the summed duration, the union of all kernels' busy time, and the idle time
inside the timed steps.  usage: python tools/trace_overlap.py <kernel_trace.csv> [skip_first_n_encode_kernels]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split("<")[0].split(" ")[-1]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id", r.get("Queue_Id", "?"))))
ev.sort()
enc = [e for e in ev if e[2] == "yk_encode2_kernel"]
t0 = enc[skip][0]; t1 = enc[-1][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
per = collections.defaultdict(float)
for s, e, n, q in win: per[n] += e - s
# union
busy = 0; cur_s, cur_e = None, None
for s, e, n, q in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
nf = len([e for e in win if e[2] == "yk_encode2_kernel"])
print(f"window {1e-6 * (t1 - t0):.3f} ms, {nf} fused kernels -> {1e-6 * (t1 - t0) / nf:.4f} ms per frame; some kernel running {100.0 * busy / (t1 - t0):.1f} % of the window")
print(f"sum of kernel durations per frame: {1e-6 * sum(per.values()) / nf:.4f} ms (overlap factor {sum(per.values()) / busy:.2f})")
for n, v in sorted(per.items(), key=lambda kv: -kv[1]):
    print(f"  {n:36s} {1e-6 * v / nf:.4f} ms per frame")
