#!/bin/bash
# usage: bash tools/abl_libs.sh "<lib1.so lib2.so ...>" "<bench args A>" "<bench args B>" ...   (dev helper: one line per (lib, args))
cd ${GRAFT_REPO_ROOT:-.}
libs=$1; shift
for lib in $libs; do
  for a in "$@"; do
    r=$(timeout -k 10 400 python tools/bench_with_lib.py $lib $a 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('step_ms=%.3f wave_ms=%.3f block_ms=%.3f merge_ms=%.3f frac=%.4f pcie_ms=%s checked=%s index_mb=%s' % (j['ms_per_step'], r['kernel_ms'], r['tier2_kernel_ms'], r['merge_kernel_ms'], r['frac'], j['config'].get('pcie_inclusive_ms_per_step'), (j.get('parity_check') or {}).get('queries'), j['config'].get('device_index_mb')))")
    echo "$lib [$a]: $r"
  done
done
