#!/bin/bash
# Dev helper (GPU box): round-3 run 34 -- dense INT8: corpus in fragment order (srx_dense_pack_i8) vs row-major; smaller sample
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3za; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_dense_int8.py -x -q -m gpu > $o/pytest_dense.log 2>&1; rc=$?; echo "pytest dense rc=$rc"; tail -3 $o/pytest_dense.log
[ $rc -eq 0 ] || exit $rc
for mode in packed rowmajor; do
  for a in "1000000 384 1024 100" "1000000 768 1024 10" "1000000 768 1024 100" "1000000 768 1024 1000" "4000000 768 1024 100" "1000000 128 1024 100" "1000000 1024 1024 100" "1000000 768 64 100" "1000000 768 256 100"; do
    echo "== $mode"; if [ $mode = rowmajor ]; then export SRX_DENSE_ROWMAJOR=1; else unset SRX_DENSE_ROWMAJOR; fi
    timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense int8\|verified"
  done
done > $o/bench_dense.log 2>&1; cat $o/bench_dense.log | grep -v verified
unset SRX_DENSE_ROWMAJOR
timeout -k 10 300 python bench.py > $o/bench_c3.log 2>&1; echo "[c3] rc=$?"; tail -1 $o/bench_c3.log | cut -c1-200
