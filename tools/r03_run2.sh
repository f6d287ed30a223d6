#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh r03b default bias0 ilp iterilp memcl 2>&1 | tee gpurun_out/r03b_ab.txt
bash tools/pmc_eval.sh r03b default
bash tools/pmc_eval.sh r03b bias0
