#!/bin/bash
# Dev helper (GPU box): round-3 run 47 -- dense INT8 filter: non-temporal loads of the B fragments (keep the query fragments in L2)
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zp; mkdir -p $o
for lib in libsparse_rx.so libsparse_rx_dnt.so libsparse_rx.so libsparse_rx_dnt.so; do
  for a in "1000000 768 1024 100" "1000000 768 1024 10" "4000000 768 1024 100" "1000000 384 1024 100" "1000000 1024 1024 100"; do
    echo "== $lib"; SRX_LIB=$lib timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense int8\|verified"
  done
done > $o/variants.log 2>&1; grep -v verified $o/variants.log; grep -c verified $o/variants.log
