#!/bin/bash
# Dev helper (GPU box): round-3 run 46 -- term-bound table of a C3-sized index: 64-bit sort (first form) vs srx_build_term_bounds
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zo; mkdir -p $o
timeout -k 10 300 python tools/bench_term_bounds.py 1e9 100000 2>&1 | grep -v Warning | grep -v amdgpu.ids | tee $o/term_bounds.log
timeout -k 10 300 python tools/bench_term_bounds.py 7.5e8 30522 2>&1 | grep -v Warning | grep -v amdgpu.ids | tee -a $o/term_bounds.log
