cd ${GRAFT_REPO_ROOT:-.}
run() { timeout -k 10 300 python bench.py --workload c2 --steps 50 --warmup 5 --no-cpu-baseline $@ 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('step_ms=%.4f wave_ms=%.4f block_ms=%.4f merge_ms=%.4f qps=%.0f' % (j['ms_per_step'], r['kernel_ms'], r['tier2_kernel_ms'], r['merge_kernel_ms'], j['value']))"; }
for a in "$@"; do echo "== $a: $(run $a)"; done
