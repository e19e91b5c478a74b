#!/bin/bash
# Dev helper (GPU box): round-3 run 37 -- 2-D layouts of the 8 GPUs (D doc shards x 8 / D query groups), rehearsed on one GPU:
# what ONE GPU does per step in each layout (C3: 10 M docs, 10 000 queries, k = 100), with the exchange among D ranks emulated on one rank
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zd; mkdir -p $o
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $o/$name.log 2>&1; echo "[$name] rc=$? $(tail -1 $o/$name.log | grep -o '"ms_per_step": [0-9.]*')"; }
run d8_q1        --docs 1250000 --queries 10000 --emulate-world 8
run d8_q1_dist   --docs 1250000 --queries 10000 --emulate-world 8 --force-dist
run d4_q2        --docs 2500000 --queries 5000  --emulate-world 4
run d4_q2_dist   --docs 2500000 --queries 5000  --emulate-world 4 --force-dist
run d2_q4        --docs 5000000 --queries 2500  --emulate-world 2
run d2_q4_dist   --docs 5000000 --queries 2500  --emulate-world 2 --force-dist
run d1_q8        --docs 10000000 --queries 1250
