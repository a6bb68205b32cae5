#!/bin/bash
# counters of every yk_* kernel of a bench stage: tools/pmc_stage.sh <tag> "<bench.py arguments>" "<CTR CTR ...>" ["<CTR ...>" ...]
# (one rocprofv3 --pmc pass per counter group; averages per dispatch and kernel; output gpurun_out/r04/pmc_stage_<tag>.txt)
TAG=$1; ARGS=$2; shift; shift
mkdir -p gpurun_out/r04
O=gpurun_out/r04/pmc_stage_$TAG.txt
: > $O
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/r04/pmcs_${TAG}_$i -- python3 bench.py $ARGS > gpurun_out/r04/pmcs_${TAG}_$i.log 2>&1 || { tail -5 gpurun_out/r04/pmcs_${TAG}_$i.log; continue; }
  python3 - "$TAG" "$i" <<'PY' >> $O
import csv, glob, collections, sys
tag, i = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(f"gpurun_out/r04/pmcs_{tag}_{i}/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].split(" ")[-1]
        if not k.startswith("yk_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k in sorted(acc):
    print(k, "dispatches", n[k], {c: round(v / n[k], 1) for c, v in acc[k].items()})
PY
  rm -rf gpurun_out/r04/pmcs_${TAG}_$i gpurun_out/r04/pmcs_${TAG}_$i.log
done
cat $O
