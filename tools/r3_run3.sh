#!/bin/bash
# Dev helper (GPU box): round-3 experiment 3 -- restart-scheme tier-1 kernel: tests + depth / clear-mode variants
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3c; mkdir -p $o
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
libs="libsparse_rx_r2w4.so libsparse_rx_r3w4.so libsparse_rx_r4w4.so libsparse_rx_r2w4a.so libsparse_rx_r3w4a.so libsparse_rx_r3w3.so libsparse_rx_r4w3.so"
bash tools/abl_libs.sh "$libs" "--no-cpu-baseline" "--docs 1250000 --emulate-world 8 --no-cpu-baseline" > $o/abl.log 2>&1; cat $o/abl.log
bash tools/abl_libs.sh "libsparse_rx_r3w4.so" "--no-cpu-baseline --debug 4" "--no-cpu-baseline --debug 1" "--no-cpu-baseline --debug 2" "--no-cpu-baseline --same-query" "--workload c2 --no-cpu-baseline" "--workload c5 --no-cpu-baseline --steps 5" "--workload c4 --no-cpu-baseline --steps 5" > $o/abl2.log 2>&1; cat $o/abl2.log
