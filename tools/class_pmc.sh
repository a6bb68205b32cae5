# per-class instruction budget of the fused kernel (run on the GPU box): classes x ablation flags
set -e
for c in frame; do for a in 0 1 3; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/cls/${c}_$a -- python3 tools/gpu_class_pmc.py $c $a > gpurun_out/cls_${c}_$a.log 2>&1
done; done
python3 - <<'PY'
import csv, glob, collections
for c in ("frame",):
  for a in (0,1,3):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/cls/{c}_{a}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "yk_encode2" in r["Kernel_Name"]:
                acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    d = list(acc.values())[-1]
    print(c, a, {k: round(v/65536,1) for k,v in d.items()})
PY
