#!/bin/bash
# per-launch durations of the 3-D LUT search kernel (six tile shapes) for one or more library builds (YK_LIB): tools/lut_phases.sh <tag> lib.so [lib.so ...]
# builds with -DYK_LUT_ABLATE=k leave phases out (see yk_lut3d.hip); output: gpurun_out/r04/lut_<tag>.txt
TAG=$1; shift
R=$(pwd); O=$R/gpurun_out/r04; mkdir -p $O
: > $O/lut_$TAG.txt
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export YK_LIB=$R/$v
  rm -rf $O/lp_$TAG
  ( cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/lp_$TAG -o run -- python3 bench.py --stage lut3d --steps 2 --warmup 1 --no-cpu --no-parity > $O/lut_${TAG}_cmd.log 2>&1 ) || { echo "rocprofv3 FAILED for $v"; tail -5 $O/lut_${TAG}_cmd.log; exit 1; }
  python3 - $O/lp_$TAG $v <<'PY' >> $O/lut_$TAG.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'lut' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
search = [r for r in rows if 'search' in r['Kernel_Name']][-6:]
other = {}
for r in rows[-40:]:
    if 'search' not in r['Kernel_Name']:
        k = r['Kernel_Name'].split('(')[0]
        other[k] = other.get(k, 0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('==', sys.argv[2])
tot = 0.0
for r in search:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print(f"  search grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):8d} wg x {r['Workgroup_Size_X']:>3s}  {d:9.1f} us")
print(f"  six passes {tot / 1e3:.3f} ms")
PY
  rm -rf $O/lp_$TAG
done
cat $O/lut_$TAG.txt
