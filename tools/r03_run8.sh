#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh r03h default trk nounc 2>&1 | tee gpurun_out/r03h_ab.txt
