#!/bin/bash
# per-kernel statistics of a command under rocprofv3 --kernel-trace --stats: tools/kstats.sh <tag> <cmd ...>  ->  gpurun_out/r04/<tag>_kernel_stats.csv (+ a short table)
TAG=$1; shift
R=$(pwd); O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ks_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$TAG -- "$@" > $O/${TAG}_cmd.out 2> $O/${TAG}_cmd.err || { echo "rocprofv3 FAILED"; tail -20 $O/${TAG}_cmd.err; exit 1; }
f=$(find $O/ks_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $O/${TAG}_kernel_stats.csv && rm -rf $O/ks_$TAG
python3 - $O/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if n.startswith("yk_") or "yk_" in n:
        print(f'{n[:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.2f} us  total {float(r["TotalDurationNs"]) / 1e6:8.3f} ms')
PY
