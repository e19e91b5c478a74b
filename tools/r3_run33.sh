#!/bin/bash
# Dev helper (GPU box): round-3 run 33 -- dense INT8: size of the threshold sample (dev build -DSRX_DENSE_KNOBS)
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3z9; mkdir -p $o
for m in 0.125 0.25 0.5 1 2 4; do
  for a in "1000000 768 1024 1000" "1000000 768 1024 100" "1000000 768 1024 10" "4000000 768 1024 100"; do
    echo "== sample x $m"; SRX_DENSE_SAMPLE_MULT=$m SRX_LIB=libsparse_rx_dknob.so timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense int8"
  done
done > $o/sample.log 2>&1; cat $o/sample.log
