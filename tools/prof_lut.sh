#!/bin/bash
# per-launch durations of the 3-D LUT search kernel (six tile shapes): tools/prof_lut.sh [size]
SZ=${1:-8192}
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r02/prof_lut
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02/prof_lut -o run -- python3 bench.py --stage lut3d --size $SZ --steps 2 --warmup 1 --no-cpu --no-parity > gpurun_out/r02/prof_lut.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/r02/prof_lut/**/*kernel_trace.csv', recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if 'lut' in r['Kernel_Name']]
for r in rows[-30:]:
    print(r['Kernel_Name'][:40], r['Grid_Size_X'], r['Workgroup_Size_X'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us')
PY
rm -rf gpurun_out/r02/prof_lut
