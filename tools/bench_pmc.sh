#!/bin/bash
# Measurement bundle (ROUND=r04 by default) of the headline workload (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats of `bench.py` (5 + 20 steps, no stage / CPU extras) -> per-kernel durations
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, L2 hit/miss) of the same program, as MI355X_MICROARCH.md prescribes
# Summaries: python tools/bench_pmc_parse.py  ->  profiles/<round>_*.json / .csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -e
R=${ROUND:-r04}
ARGS="--no-stages --no-cpu-baseline --no-alt --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_kstats -- python bench.py --steps 20 --warmup 5 $ARGS > gpurun_out/${R}_kstats.log 2>&1
echo kstats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_pmc_fetch -- python bench.py --steps 3 --warmup 1 $ARGS > gpurun_out/${R}_pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_pmc_write -- python bench.py --steps 3 --warmup 1 $ARGS > gpurun_out/${R}_pmc_write.log 2>&1
echo write done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${R}_pmc_l2 -- python bench.py --steps 3 --warmup 1 $ARGS > gpurun_out/${R}_pmc_l2.log 2>&1
echo l2 done
