#!/bin/bash
# Dev helper (GPU box): round-3 run 29 -- dense INT8 filter: ping-pong form (two groups of four waves half a tile apart)
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3z5; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_dense_int8.py -x -q -m gpu > $o/pytest_dense.log 2>&1; rc=$?; echo "pytest dense rc=$rc"; tail -3 $o/pytest_dense.log
[ $rc -eq 0 ] || exit $rc
for lib in libsparse_rx.so libsparse_rx_dpp99.so libsparse_rx_dpp1.so; do
  for a in "1000000 384 1024 100" "1000000 768 1024 10" "1000000 768 1024 100" "4000000 768 1024 100" "1000000 128 1024 100" "1000000 1024 1024 100" "1000000 512 1024 100" "1000000 192 1024 100"; do
    echo "== $lib"; SRX_LIB=$lib timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense int8"
  done
done > $o/variants.log 2>&1; cat $o/variants.log
for a in "1000000 768 1024 100" "1000000 384 1024 100"; do
  echo "== stamps $a"; SRX_LIB=libsparse_rx_dstamp.so timeout -k 10 200 python tools/dense_stamp_run.py $a 2>&1 | grep -v Warning | grep -v amdgpu.ids
done > $o/stamps.log 2>&1; cat $o/stamps.log
