#!/bin/bash
bash tools/gpu_suite.sh || exit 1
for sz in 2048 8192; do
  timeout -k 10 300 python bench.py --stage lut3d --size $sz --steps 3 --warmup 1 > gpurun_out/r02/bench_stage_lut3d_$sz.json 2> gpurun_out/r02/bench_stage_lut3d_$sz.err; echo "lut3d $sz rc=$?"; cat gpurun_out/r02/bench_stage_lut3d_$sz.json | cut -c1-900; tail -3 gpurun_out/r02/bench_stage_lut3d_$sz.err
done
