# per-class instruction budget of the fused kernel (run on the GPU box): classes x ablation flags
#   usage: tools/class_pmc.sh "frame ramp mild noise" "0 1"
set -e
CLS=${1:-"frame ramp mild noise"}; ABL=${2:-"0 1"}
for c in $CLS; do for a in $ABL; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/cls/${c}_$a -- python3 tools/gpu_class_pmc.py $c $a > gpurun_out/cls_${c}_$a.log 2>&1
done; done
CLS="$CLS" ABL="$ABL" python3 - <<'PY'
import csv, glob, collections, os
for c in os.environ["CLS"].split():
  for a in os.environ["ABL"].split():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/cls/{c}_{a}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "yk_encode2" in r["Kernel_Name"]:
                acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    d = list(acc.values())[-1]
    print(c, a, {k: round(v/65536,1) for k,v in d.items()})
PY
