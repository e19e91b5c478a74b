#!/bin/bash
# Dev helper (GPU box): round-3 run 45 -- last smoke of the side legs: self-launching bench, multi-stream steady state, one-rank RCCL exchange
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zn; mkdir -p $o
timeout -k 10 300 python bench.py --workload c2 --gpus 1 --self-launch > $o/selflaunch_c2.log 2>&1; echo "[selflaunch c2] rc=$? $(tail -1 $o/selflaunch_c2.log | grep -o '"ms_per_step": [0-9.]*')"
timeout -k 10 300 python bench.py --workload c2 --streams 4 > $o/c2_steady.log 2>&1; echo "[c2 steady] rc=$? $(tail -1 $o/c2_steady.log | grep -o '"steady_state": {[^}]*}' | cut -c1-300)"
timeout -k 10 300 python bench.py --docs 1250000 --emulate-world 8 --force-dist > $o/shard_dist.log 2>&1; echo "[shard dist] rc=$? $(tail -1 $o/shard_dist.log | grep -o '"ms_per_step": [0-9.]*')"
