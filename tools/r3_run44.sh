#!/bin/bash
# Dev helper (GPU box): round-3 run 44 -- kernel timeline of the dense INT8 search at k = 10
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zl; mkdir -p $o
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $o/d768k10 -o dense -- python3 tools/bench_dense.py 1000000 768 1024 10 > $o/prof.log 2>&1; echo "rc=$?"
ls -la $o/d768k10
