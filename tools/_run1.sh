cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "splade or zipf or fuzz or uniform" > gpurun_out/r2j/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r2j/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload c4 --no-cpu-baseline > gpurun_out/r2j/bench_c4.log 2>&1; rc=$?; echo "c4 rc=$rc"; tail -1 gpurun_out/r2j/bench_c4.log | cut -c1-200; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload c4 > gpurun_out/r2j/bench_c4_chk.log 2>&1; rc=$?; echo "c4 checked rc=$rc"; tail -1 gpurun_out/r2j/bench_c4_chk.log | cut -c1-200; exit $rc
