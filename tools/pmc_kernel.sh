#!/bin/bash
# Generic PMC study of one kernel: tools/pmc_kernel.sh <tag> <kernel-name-substring> <python script> [args...]
# One rocprofv3 --pmc pass per counter group (never combined with tracing), then per-launch means of the matching kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; SUB=$2; shift 2
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  d=gpurun_out/pmc_${TAG}_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 "$@" > $d.log 2>&1 || echo "pass failed: $c"
done
python3 - "$TAG" "$SUB" <<'PY'
import csv, glob, json, sys
tag, sub = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True)):
    per = {}
    for row in csv.DictReader(open(f)):
        if sub not in row["Kernel_Name"]: continue
        per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for ctr, disp in per.items():
        v = list(disp.values())
        res[ctr] = {"per_launch_mean": sum(v) / len(v), "launches": len(v), "first": v[0]}
json.dump(res, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
for k, v in res.items():
    print(k, round(v["per_launch_mean"]), "x", v["launches"])
PY
