#!/bin/bash
# timeline of the kernels of a few steady-state frames of a command: tools/ktimeline.sh <tag> <cmd ...>  (rocprofv3 --kernel-trace)
TAG=$1; shift
R=$(pwd); O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt_$TAG
rocprofv3 --kernel-trace --output-format csv -d $O/kt_$TAG -- "$@" > $O/${TAG}_cmd.out 2> $O/${TAG}_cmd.err || { echo "rocprofv3 FAILED"; tail -20 $O/${TAG}_cmd.err; exit 1; }
f=$(find $O/kt_$TAG -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY' > $O/${TAG}_timeline.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "yk_" in r["Kernel_Name"] and "roof" not in r["Kernel_Name"] and "qtab" not in r["Kernel_Name"] and "deftab" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fused = [i for i, r in enumerate(rows) if "encode2" in r["Kernel_Name"]]
if len(fused) > 12:
    a, b = fused[len(fused) // 2], fused[len(fused) // 2 + 5]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a - 6:b + 1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f'{r["Kernel_Name"].split("(")[0][:28]:28s} q{r.get("Queue_Id", "?"):>3s} start {s / 1e3:9.1f} us  end {e / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}')
PY
rm -rf $O/kt_$TAG
cat $O/${TAG}_timeline.txt
