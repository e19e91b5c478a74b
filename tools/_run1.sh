cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "splade or zipf or fuzz or uniform or edge or text" > gpurun_out/r2j/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r2j/pytest.log; [ $rc -eq 0 ] || exit $rc
bash tools/abl_libs.sh "libsparse_rx.so" "--workload c4 --no-cpu-baseline --steps 5" "--workload c5 --no-cpu-baseline --steps 5" "--workload c1 --no-cpu-baseline" "--workload c3 --no-cpu-baseline" "--workload c4 --no-cpu-baseline --steps 5 --debug 16384" > gpurun_out/r2j/abl_c4c.log 2>&1; cat gpurun_out/r2j/abl_c4c.log
