#!/bin/bash
# HBM traffic of the head conv (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python tools/conv_only.py 32 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python tools/conv_only.py 32 > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_l2 -- python tools/conv_only.py 32 > gpurun_out/pmc_l2.log 2>&1
ls gpurun_out/pmc_fetch/*/ gpurun_out/pmc_write/*/
