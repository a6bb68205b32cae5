#!/bin/bash
# round 2, experiment 1: cache policy of the quantiser-table gathers + TA / TCP counters of the fused kernel
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp1.log
: > $O
for v in build/libBase.so build/libQ1.so build/libQ3.so build/libQ17.so build/libBase.so; do
  cp $v yaik_amd/libyaik_hip.so
  echo "== $v" >> $O
  timeout -k 10 120 python tools/gpu_class_cost.py 2>&1 | grep "mode3=0" >> $O
  timeout -k 10 120 python tools/gpu_class_pmc.py frame 0 2>&1 | grep "fused kernel" >> $O
done
cp build/libBase.so yaik_amd/libyaik_hip.so
pmc() { rocprofv3 --pmc $3 --output-format csv -d gpurun_out/r02/pmc_$1_$2 -- python3 tools/gpu_class_pmc.py $1 0 > gpurun_out/r02/pmc_$1_$2.log 2>&1; }
for cls in noise frame mild; do
  pmc $cls a "TA_TA_BUSY_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
  pmc $cls b "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum"
  pmc $cls c "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"
done
python3 - <<'PY' >> gpurun_out/r02/exp1.log
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
cat $O
