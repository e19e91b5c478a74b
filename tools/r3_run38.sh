#!/bin/bash
# Dev helper (GPU box): round-3 run 38 -- final dense INT8 evidence: rocprofv3 kernel stats + PMC passes + stamps of the shipped form
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3ze; mkdir -p $o
bash tools/dense_profile.sh d768final 1000000 768 1024 100 > gpurun_out/prof_d768final.log 2>&1; echo "d768 rc=$?"
for a in "1000000 768 1024 100" "1000000 384 1024 100"; do
  echo "== stamps $a"; SRX_LIB=libsparse_rx_dstamp.so timeout -k 10 200 python tools/dense_stamp_run.py $a 2>&1 | grep -v Warning | grep -v amdgpu.ids
done > $o/stamps.log 2>&1; cat $o/stamps.log
