#!/bin/bash
# rocprofv3 kernel stats of the default bench command (100 steps): the average of yk_encode2_kernel must agree with the line's kernel_ms
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -o run -- python3 bench.py --no-cpu --no-parity > gpurun_out/final/bench_under_tracer.json 2> gpurun_out/final/stats.err
echo "rc=$?"
python3 - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/final/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'yk_' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
d = json.loads(open('gpurun_out/final/bench_under_tracer.json').readline()); print(d['value'], d['roofline']['kernel_ms'])
PY
