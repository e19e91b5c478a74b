#!/bin/bash
# Dev helper (GPU box): round-3 baseline of the shard step and its exchange rehearsal
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3a; mkdir -p $o
timeout -k 10 300 python bench.py --docs 1250000 --emulate-world 8 --no-cpu-baseline > $o/shard_emulate8.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --docs 1250000 --emulate-world 8 --no-cpu-baseline --force-dist > $o/shard_dist_a2a.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --docs 1250000 --emulate-world 8 --no-cpu-baseline --force-dist --no-overlap > $o/shard_dist_noov.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $o/c3.log 2>&1 || exit 1
grep -h "host submit\|ms_per_step" $o/*.log | cut -c1-400
