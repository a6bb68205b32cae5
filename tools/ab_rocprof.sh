#!/bin/bash
# same-box A/B of library builds by the rocprofv3 average of the fused kernel over the default bench command (210 launches each, twice):
# tools/ab_rocprof.sh <tag> libA libB ...
TAG=$1; shift
mkdir -p gpurun_out/r02
O=$GRAFT_REPO_ROOT/gpurun_out/r02/abr_$TAG.log; : > $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  cp $v yaik_amd/libyaik_hip.so
  rm -rf /tmp/abr_stats
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abr_stats -o run -- python3 bench.py --no-cpu --no-parity > /tmp/abr_bench.json 2> /dev/null
  python3 - "$v" >> $O <<'PY'
import csv, glob, json, sys
f = glob.glob('/tmp/abr_stats/**/*kernel_stats.csv', recursive=True)[0]
avg = [r for r in csv.DictReader(open(f)) if 'yk_encode2_kernel' in r['Name']][0]
d = json.loads(open('/tmp/abr_bench.json').readline())
print(sys.argv[1], 'rocprof avg us', round(float(avg['AverageNs']) / 1e3, 2), 'calls', avg['Calls'], 'bench', d['value'], d['roofline']['kernel_ms'])
PY
done
done
cat $O
