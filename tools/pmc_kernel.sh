#!/bin/bash
# counters of one kernel of any command: tools/pmc_kernel.sh <tag> <kernel name substring> "<CTR CTR ...>" ["<CTR ...>" ...] -- <program> <args...>
# one rocprofv3 --pmc pass per counter group (at most two TA/TCP counters per pass on gfx950); the LAST dispatch of the kernel is printed.
TAG=$1; KERN=$2; shift; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
mkdir -p gpurun_out/r04
O=gpurun_out/r04/pmck_$TAG.txt
: > $O
i=0
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/r04/pmck_${TAG}_$i -- "$@" > gpurun_out/r04/pmck_${TAG}_$i.log 2>&1 || { echo "rocprofv3 pass $i ($grp) FAILED:"; tail -25 gpurun_out/r04/pmck_${TAG}_$i.log; exit 1; }
  python3 - "$TAG" "$i" "$KERN" <<'PY' >> $O
import csv, glob, collections, sys
tag, i, kern = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/r04/pmck_{tag}_{i}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
if acc:
    d = acc[max(acc)]
    print({k: round(v, 1) for k, v in d.items()})
PY
  rm -rf gpurun_out/r04/pmck_${TAG}_$i gpurun_out/r04/pmck_${TAG}_$i.log
done
cat $O
