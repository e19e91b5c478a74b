#!/bin/bash
# Dev helper (GPU box): round-3 experiment 2 -- GPU test suite on the new build, tier-1 depth / occupancy variants, stream microbench
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3b; mkdir -p $o
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
libs="libsparse_rx_d2w4.so libsparse_rx_d3w4.so libsparse_rx_d2w3.so libsparse_rx_d3w3.so libsparse_rx_d4w3.so libsparse_rx_d3w3L384.so libsparse_rx_d4w3L384.so"
bash tools/abl_libs.sh "$libs" "--no-cpu-baseline" "--docs 1250000 --emulate-world 8 --no-cpu-baseline" > $o/abl.log 2>&1; cat $o/abl.log
timeout -k 10 120 tools/stream_microbench2 10000 205 > $o/micro_205.log 2>&1; cat $o/micro_205.log
timeout -k 10 120 tools/stream_microbench2 10000 26 > $o/micro_26.log 2>&1; cat $o/micro_26.log
timeout -k 10 300 python bench.py --gpus 1 --self-launch --docs 1250000 --steps 5 --no-cpu-baseline > $o/selflaunch.log 2>&1; echo "selflaunch rc=$?"; tail -2 $o/selflaunch.log | cut -c1-300
