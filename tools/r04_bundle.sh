#!/bin/bash
# Round-4 measurement bundle (one gpurun call): kernel stats + PMC passes of the headline, the MLP-branch PMC study, kernel stats of
# the cfg0 / cfg3 / LoftUp workloads, and a full bench line.  Summaries land in profiles/ (copied back through gpurun_out/profiles_copy).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ROUND=r04
bash tools/bench_pmc.sh && python3 tools/bench_pmc_parse.py > gpurun_out/r04_pmc_parse.log 2>&1
bash tools/mlp_pmc.sh > gpurun_out/r04_mlp_pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_cfg0_kstats -- python3 tools/cfg0_only.py > gpurun_out/r04_cfg0.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_cfg3_kstats -- python3 tools/bench_cfg3.py > gpurun_out/r04_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_loftup_kstats -- python3 tools/bench_loftup.py 8 > gpurun_out/r04_loftup.log 2>&1
python3 - <<PY
import glob, shutil
for tag, name in (("r04_cfg0_kstats", "r04_cfg0_bilinear448_b32_kernel_stats.csv"), ("r04_cfg3_kstats", "r04_cfg3_vitl14_lift896_kernel_stats.csv"),
                  ("r04_loftup_kstats", "r04_loftup448_b8_kernel_stats.csv")):
    f = glob.glob(f"gpurun_out/{tag}/**/*kernel_stats.csv", recursive=True)
    if f:
        shutil.copy(f[0], "profiles/" + name)
PY
python3 bench.py > profiles/r04_bench_line.json 2> gpurun_out/r04_bench_line.err
mkdir -p gpurun_out/profiles_copy && cp profiles/r04_* gpurun_out/profiles_copy/
tail -3 gpurun_out/r04_pmc_parse.log; tail -8 gpurun_out/r04_mlp_pmc.log; tail -1 gpurun_out/r04_cfg0.log gpurun_out/r04_cfg3.log gpurun_out/r04_loftup.log
python3 -c "
import json; d=json.load(open('profiles/r04_bench_line.json'))
print('headline', d['value'], d['ms_per_step'], 'conv frac', d['roofline']['frac'], 'traffic/alg', d['roofline'].get('traffic_over_algorithmic'))
print('vit', d['roofline_vit']['frac'], d['roofline_vit']['frac_in_step'], 'ups', d['roofline_upsampler']['frac'], d['roofline_upsampler']['ms_per_step'])
for k in ('cfg0_bilinear448','cfg3_vitl14_lift896','loftup448','size896','cfg2_train_vits14_loftup224','fp32_mode'):
    v=d.get(k,{}); print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if not isinstance(b,(dict,str))})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['batch8_images_per_sec'], d['cpu_baseline']['cores'], d['cpu_baseline']['os_cpu_count'])
"
