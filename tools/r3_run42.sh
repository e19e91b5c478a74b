#!/bin/bash
# Dev helper (GPU box): round-3 run 42 -- C4: blocks in flight per lane in the wave-level dense tiles (3 / 4 = shipped / 6)
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zj; mkdir -p $o
bash tools/abl_libs.sh "libsparse_rx.so libsparse_rx_wd3.so libsparse_rx_wd6.so libsparse_rx.so" "--workload c4 --no-cpu-baseline --steps 10" > $o/abl.log 2>&1; cat $o/abl.log
