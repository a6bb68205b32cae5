# where do the waves of the fused kernel wait? (run on the GPU box; separate passes keep each counter set small)
set -e
CLS=${1:-frame}
run() { rocprofv3 --pmc $2 --output-format csv -d gpurun_out/stall/$1 -- python3 tools/gpu_class_pmc.py $CLS 0 > gpurun_out/stall_$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
run b "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
run c "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU"
python3 - <<'PY'
import csv, glob, collections
for tag in "abc":
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/stall/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "yk_encode2" in r["Kernel_Name"]:
                acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    d = list(acc.values())[-1]
    print({k: round(v/65536,1) for k,v in d.items()})
PY
