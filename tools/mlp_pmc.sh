#!/bin/bash
# PMC passes for the ViT MLP branch (fused kernel vs the three-kernel route): each pass its own run, --pmc only.
# Writes profiles/${ROUND:-r04}_vit_mlp_pmc.json (per kernel and counter: mean per launch).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${ROUND:-r04}
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  d=gpurun_out/pmc_mlp_$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/mlp_pmc_run.py > $d.log 2>&1 || echo "pass failed: $c"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_mlp_stats -- python3 tools/mlp_pmc_run.py > gpurun_out/pmc_mlp_stats.log 2>&1
python3 - <<PY
import csv, glob, json
res = {}
for f in sorted(glob.glob("gpurun_out/pmc_mlp_*/**/*counter_collection.csv", recursive=True)):
    per = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(s in k for s in ("vit_mlp_fused", "gemm_tile_kernel", "layernorm_kernel")): continue
        name = "vit_mlp_fused" if "vit_mlp_fused" in k else ("layernorm" if "layernorm" in k else ("gemm_fc1_gelu" if "Gelu" in k or "GELU" in k or "Act" in k else "gemm_" + str(abs(hash(k)) % 1000)))
        per.setdefault((name, k[:160], row["Counter_Name"]), {}).setdefault(row["Dispatch_Id"], 0.0)
        per[(name, k[:160], row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (name, full, ctr), disp in per.items():
        v = list(disp.values())
        res.setdefault(name, {"kernel": full})[ctr] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
for f in glob.glob("gpurun_out/pmc_mlp_stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for name, d in res.items():
            if d["kernel"][:120] == row["Name"][:120]:
                d["avg_us_unprofiled_pass"] = float(row["AverageNs"]) / 1e3
json.dump(res, open("profiles/${R}_vit_mlp_pmc.json", "w"), indent=1)
for name, d in res.items():
    print(name, {k: (round(v["per_launch_mean"]) if isinstance(v, dict) else v) for k, v in d.items() if k != "kernel"})
PY
mkdir -p gpurun_out/profiles_copy && cp profiles/${R}_vit_mlp_pmc.json gpurun_out/profiles_copy/
