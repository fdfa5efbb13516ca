#!/bin/bash
# Round-2 measurement bundle of the headline workload (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats of `bench.py` (5 + 20 steps, no stage / CPU extras) -> per-kernel durations
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, L2 hit/miss) of the same program, as MI355X_MICROARCH.md prescribes
# Summaries: python tools/bench_pmc_parse.py  ->  profiles/r02_*.json / .csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -e
ARGS="--no-stages --no-cpu-baseline --no-alt"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_kstats -- python bench.py --steps 20 --warmup 5 $ARGS > gpurun_out/r02_kstats.log 2>&1
echo kstats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02_pmc_fetch -- python bench.py --steps 3 --warmup 1 $ARGS > gpurun_out/r02_pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02_pmc_write -- python bench.py --steps 3 --warmup 1 $ARGS > gpurun_out/r02_pmc_write.log 2>&1
echo write done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r02_pmc_l2 -- python bench.py --steps 3 --warmup 1 $ARGS > gpurun_out/r02_pmc_l2.log 2>&1
echo l2 done
