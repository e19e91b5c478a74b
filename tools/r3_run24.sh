#!/bin/bash
# Dev helper (GPU box): round-3 run 24 -- dense INT8 filter: in-kernel stamps (8 and 4 waves per workgroup), variants
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3x; mkdir -p $o
for lib in libsparse_rx_dstamp.so libsparse_rx_dstamp4.so; do
  for a in "1000000 768 1024 100" "1000000 128 1024 100"; do
    echo "== $lib $a"; SRX_LIB=$lib timeout -k 10 200 python tools/dense_stamp_run.py $a 2>&1 | grep -v Warning | grep -v amdgpu.ids
  done
done > $o/stamps.log 2>&1; cat $o/stamps.log
for lib in libsparse_rx.so libsparse_rx_dnw4.so libsparse_rx_dpf2.so; do
  for a in "1000000 768 1024 100" "4000000 768 1024 100" "1000000 1024 1024 100"; do
    echo "== $lib"; SRX_LIB=$lib timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense int8"
  done
done > $o/variants.log 2>&1; cat $o/variants.log
