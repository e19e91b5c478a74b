#!/bin/bash
# Dev helper (GPU box): round-3 run 41 -- dense INT8 filter: s_setprio 1 / 3 around the MFMA loop
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zh; mkdir -p $o
for lib in libsparse_rx.so libsparse_rx_dprio1.so libsparse_rx_dprio3.so; do
  for a in "1000000 384 1024 100" "1000000 768 1024 10" "1000000 768 1024 100" "4000000 768 1024 100" "1000000 128 1024 100" "1000000 1024 1024 100"; do
    echo "== $lib"; SRX_LIB=$lib timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense int8\|verified"
  done
done > $o/variants.log 2>&1; grep -v verified $o/variants.log; grep -c verified $o/variants.log
