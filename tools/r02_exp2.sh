#!/bin/bash
# round 2, run 2: op rates, the new parity tests, TA / TCP counters (few counters per pass)
mkdir -p gpurun_out/r02
cp build/libBase.so /tmp/libBase.so 2>/dev/null
timeout -k 10 300 tools/ubench/op_rate > gpurun_out/r02/op_rate.txt 2>&1
tail -16 gpurun_out/r02/op_rate.txt
timeout -k 10 900 python -m pytest tests/test_gpu_decode_parity.py tests/test_gpu_host_chunks.py tests/test_gpu_host_mirror.py -x -q > gpurun_out/r02/pytest_a.log 2>&1; tail -3 gpurun_out/r02/pytest_a.log
timeout -k 10 900 python -m pytest tests/test_gpu_encode_parity.py -x -q -k "non_synthetic or reference_hashes" > gpurun_out/r02/pytest_b.log 2>&1; tail -3 gpurun_out/r02/pytest_b.log
timeout -k 10 900 python -m pytest tests/test_gpu_baseline_configs.py -x -q -k "headline" > gpurun_out/r02/pytest_c.log 2>&1; tail -3 gpurun_out/r02/pytest_c.log
pmc() { timeout -k 10 150 rocprofv3 --pmc $3 --output-format csv -d gpurun_out/r02/pmc_$1_$2 -- python3 tools/gpu_class_pmc.py $1 0 > gpurun_out/r02/pmc_$1_$2.log 2>&1; echo "pmc $1 $2 rc=$?"; }
for cls in noise frame; do
  pmc $cls a "TA_TA_BUSY_sum TA_BUSY_avr"
  pmc $cls b "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
  pmc $cls c "TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
  pmc $cls d "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
  pmc $cls e "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU"
done
python3 - <<'PY' > gpurun_out/r02/exp2_pmc.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r02/pmc_*_?")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "yk_encode2" in r["Kernel_Name"]:
                acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if acc:
        last = list(acc.values())[-1]
        print(d, {k: round(v, 1) for k, v in last.items()})
PY
cat gpurun_out/r02/exp2_pmc.txt
