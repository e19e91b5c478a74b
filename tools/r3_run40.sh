#!/bin/bash
# Dev helper (GPU box): round-3 run 40 -- the default bench line on the committed tree (traffic.json matches the kernel sources)
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zg; mkdir -p $o
timeout -k 10 300 python bench.py > $o/bench_c3.log 2>&1; echo "[c3] rc=$?"; tail -1 $o/bench_c3.log | cut -c1-400
