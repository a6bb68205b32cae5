#!/bin/bash
# rocprofv3 kernel stats of one stage bench: tools/prof_stage.sh <stage>
ST=$1
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_$ST -o run -- python3 bench.py --stage $ST --steps 10 --warmup 2 > gpurun_out/r02/prof_$ST.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/r02/prof_$ST/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows:
    if r['Name'].startswith('yk_') or 'rocclr' in r['Name']:
        print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'])
PY
