#!/bin/bash
# end-of-round evidence: smoke, the whole GPU suite, the driver-style bench line and the stage lines (outputs under gpurun_out/final/)
mkdir -p gpurun_out/final
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/final/smoke.log
bash tools/gpu_suite.sh || exit 1
timeout -k 10 400 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/final/bench_default.json
for st in corners range1d decode; do
  timeout -k 10 300 python bench.py --stage $st --steps 10 --warmup 2 > gpurun_out/final/bench_stage_$st.json 2> gpurun_out/final/bench_stage_$st.err; echo "stage $st rc=$?"
done
timeout -k 10 300 python bench.py --layout stripes --steps 10 --warmup 2 > gpurun_out/final/bench_stripes_n1.json 2> gpurun_out/final/bench_stripes_n1.err; echo "stripes rc=$?"
