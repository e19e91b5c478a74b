#!/bin/bash
# Dev helper (GPU box): round-3 run 12 -- what the search-after bound test and a deeper block ring cost / gain in tier 2
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3l; mkdir -p $o
bash tools/abl_libs.sh "libsparse_rx.so libsparse_rx_noafter.so libsparse_rx_wd8.so libsparse_rx.so libsparse_rx_noafter.so" "--workload c4 --no-cpu-baseline --steps 10" "--workload c5 --no-cpu-baseline --steps 10" > $o/abl.log 2>&1; cat $o/abl.log
