#!/bin/bash
# Dev helper (GPU box): round-3 run 16 -- work-item plans (target blocks per round) with the in-kernel merge
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3p; mkdir -p $o
S="--docs 1250000 --emulate-world 8 --no-cpu-baseline"
bash tools/abl_libs.sh "libsparse_rx.so" "$S" "$S --target-blocks 4096" "$S --target-blocks 4864" "$S --target-blocks 5120" "$S --target-blocks 6144" "$S --target-blocks 9728" "--no-cpu-baseline" "--no-cpu-baseline --target-blocks 4864" "--no-cpu-baseline --target-blocks 5120" "--no-cpu-baseline --target-blocks 9728" "--workload c2 --no-cpu-baseline" "--workload c2 --no-cpu-baseline --target-blocks 4864" "--workload c2 --no-cpu-baseline --target-blocks 2048" > $o/abl.log 2>&1; cat $o/abl.log
