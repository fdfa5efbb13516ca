#!/bin/bash
# PMC passes for the ViT self-attention kernel (B=32, L=1025, 6 heads): each pass its own run, --pmc only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC"; do
  d=gpurun_out/pmc_att_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python tools/att_bench.py 32 1025 6 5 > $d.log 2>&1 || echo "pass failed: $c"
done
python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/pmc_att_*/**/*counter_collection.csv", recursive=True)):
    per = {}
    for row in csv.DictReader(open(f)):
        if "attention" not in row["Kernel_Name"]: continue
        per.setdefault((row["Kernel_Name"][:60], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for k, v in per.items():
        print(f"{k[0]} {k[1]}: mean per launch {sum(v)/len(v):.4g} over {len(v)} launches")
PY
