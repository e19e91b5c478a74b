#!/bin/bash
# Dev helper (runs on the GPU box through gpurun): GPU parity tests, then bench ablations of the scoring kernels.
# usage: bash tools/gpu_ablate.sh <tag> [test|notest] "<bench args 1>" "<bench args 2>" ...
cd ${GRAFT_REPO_ROOT:-.}
tag=$1; shift
dotest=$1; shift
if [ "$dotest" = "test" ]; then
  timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_$tag.log 2>&1; echo rc=$? >> gpurun_out/gputest_$tag.log; tail -5 gpurun_out/gputest_$tag.log
fi
run() { timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $@ 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=j['roofline']; print('step_ms=%.3f wave_ms=%.3f block_ms=%.3f merge_ms=%.3f frac=%.4f' % (j['ms_per_step'], r['kernel_ms'], r['tier2_kernel_ms'], r['merge_kernel_ms'], r['frac']))"; }
rm -f gpurun_out/abl_$tag.log
for a in "$@"; do echo "== $a: $(run $a)" >> gpurun_out/abl_$tag.log; done
cat gpurun_out/abl_$tag.log
