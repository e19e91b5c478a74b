#!/bin/bash
# Dev helper (GPU box): round-3 run 39 -- dense INT8 filter: workgroups start at different query tiles (stagger 5; 0 = in step; 13)
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zf; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_dense_int8.py -x -q -m gpu > $o/pytest_dense.log 2>&1; rc=$?; echo "pytest dense rc=$rc"; tail -3 $o/pytest_dense.log
[ $rc -eq 0 ] || exit $rc
for lib in libsparse_rx.so libsparse_rx_dstag0.so libsparse_rx_dstag13.so; do
  for a in "1000000 384 1024 100" "1000000 768 1024 10" "1000000 768 1024 100" "4000000 768 1024 100" "1000000 128 1024 100" "1000000 1024 1024 100" "1000000 768 256 100"; do
    echo "== $lib"; SRX_LIB=$lib timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense int8"
  done
done > $o/variants.log 2>&1; cat $o/variants.log
