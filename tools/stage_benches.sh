#!/bin/bash
# the bench lines of every stage (driver-style invocations) + rocprofv3 kernel stats of the other stages
mkdir -p gpurun_out/r02
for st in corners range1d decode; do
  timeout -k 10 300 python bench.py --stage $st --steps 10 --warmup 2 > gpurun_out/r02/bench_stage_$st.json 2> gpurun_out/r02/bench_stage_$st.err; echo "stage $st rc=$?"; cat gpurun_out/r02/bench_stage_$st.json
done
timeout -k 10 300 python bench.py --layout stripes --steps 10 --warmup 2 > gpurun_out/r02/bench_stripes_n1.json 2> gpurun_out/r02/bench_stripes_n1.err; echo "stripes n1 rc=$?"; cat gpurun_out/r02/bench_stripes_n1.json
# N = 2 rehearsal of the stripes layout on the one GPU of this box: gloo, both ranks on GPU 0 (plumbing only, not a scaling number)
YK_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --layout stripes --steps 4 --warmup 1 > gpurun_out/r02/bench_stripes_gloo2.json 2> gpurun_out/r02/bench_stripes_gloo2.err; echo "stripes gloo2 rc=$?"; tail -1 gpurun_out/r02/bench_stripes_gloo2.json
