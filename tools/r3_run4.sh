#!/bin/bash
# Dev helper (GPU box): round-3 experiment 4 -- lean pass 1 at depth 2 / 3, work-item plans without a split tail, new tests
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3d; mkdir -p $o
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
S="--docs 1250000 --emulate-world 8 --no-cpu-baseline"
bash tools/abl_libs.sh "libsparse_rx_k2w4.so libsparse_rx_k3w4.so" "--no-cpu-baseline" "$S" "--no-cpu-baseline --target-blocks 10000" "$S --target-blocks 10000" "--no-cpu-baseline --target-blocks 4096" "$S --target-blocks 4096" "$S --target-blocks 2048" > $o/abl.log 2>&1; cat $o/abl.log
timeout -k 10 300 python bench.py --docs 1250000 --emulate-world 8 > $o/shard_emu_check.log 2>&1; echo "emu check rc=$?"; grep -h "PARITY\|parity_check" $o/shard_emu_check.log | cut -c1-300 | tail -2
