#!/bin/bash
# PMC passes for jbu_kernels at the 512^2 stage (each pass its own run, --pmc only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  d=gpurun_out/pmc_jbuk_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python tools/jbu_kernels_only.py 32 > $d.log 2>&1
done
python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/pmc_jbuk_*/**/*counter_collection.csv", recursive=True)):
    per = {}
    for row in csv.DictReader(open(f)):
        if "jbu_kernels_kernel" not in row["Kernel_Name"]: continue
        per.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in per.items():
        print(f"{k}: mean per launch {sum(v)/len(v):.4g} over {len(v)} launches")
PY
