#!/bin/bash
# usage: bash tools/gpu_ablate_lib.sh <tag> <lib.so> "<bench args 1>" ...
cd ${GRAFT_REPO_ROOT:-.}
tag=$1; shift
lib=$1; shift
run() { timeout -k 10 300 python tools/bench_with_lib.py $lib --steps 5 --warmup 1 --no-cpu-baseline $@ 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=j['roofline']; print('step_ms=%.3f wave_ms=%.3f block_ms=%.3f merge_ms=%.3f frac=%.4f' % (j['ms_per_step'], r['kernel_ms'], r['tier2_kernel_ms'], r['merge_kernel_ms'], r['frac']))"; }
rm -f gpurun_out/abl_$tag.log
for a in "$@"; do echo "== $a: $(run $a)" >> gpurun_out/abl_$tag.log; done
cat gpurun_out/abl_$tag.log
