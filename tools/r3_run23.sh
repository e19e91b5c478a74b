#!/bin/bash
# Dev helper (GPU box): round-3 run 23 -- PMC profile of the dense INT8 path (1 M x 768, 1 024 queries, k = 100; and dim 128)
cd ${GRAFT_REPO_ROOT:-.}
bash tools/dense_profile.sh d768 1000000 768 1024 100 > gpurun_out/prof_d768.log 2>&1; echo "d768 rc=$?"
bash tools/dense_profile.sh d128 1000000 128 1024 100 > gpurun_out/prof_d128.log 2>&1; echo "d128 rc=$?"
tail -5 gpurun_out/prof_d768.log
