"""Per-kernel average durations of the decode stage from a rocprofv3 --kernel-trace --stats directory.  usage: python tools/decode_kernel_stats.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    if "yk_dec" in n or "rocclr" in n:
        print("%-52s calls %4s avg %8.1f us" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
