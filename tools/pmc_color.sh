#!/bin/bash
# usage (GPU box): tools/pmc_color.sh  -- MFMA-busy / LDS / wait counters of the colour kernels (tools/color_bench.py)
export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  name=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_col_$name -o c -- python3 tools/color_bench.py > gpurun_out/pmc_col_$name.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_col_$name/**/*counter_collection.csv",recursive=True)
if not f: print("no counters for $set"); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"][:45]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "color_" in k or "geo_pairs" in k:
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
