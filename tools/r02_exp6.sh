#!/bin/bash
# GPU suite, then the stage bench lines (corners / decode after their rewrites)
bash tools/gpu_suite.sh || exit 1
for st in corners decode range1d; do
  timeout -k 10 300 python bench.py --stage $st --steps 10 --warmup 2 > gpurun_out/r02/bench_stage_$st.json 2> gpurun_out/r02/bench_stage_$st.err; echo "stage $st rc=$?"; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r02/bench_stage_$st.json').readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline'].get('note'))"
done
