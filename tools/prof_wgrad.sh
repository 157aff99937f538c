#!/bin/bash
# usage: tools/prof_wgrad.sh   (on the GPU box) -- kernel durations and MFMA-busy counters of tools/wgrad_bench2.py
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wg -o wg -- python3 tools/wgrad_bench2.py > gpurun_out/prof_wg.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_wg/**/*kernel_stats.csv",recursive=True)
print(f)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_wg -o wg -- python3 tools/wgrad_bench2.py > gpurun_out/pmc_wg.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_wg/**/*counter_collection.csv",recursive=True)
print(f)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "wgrad" in k or "Cijk" in k:
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
