cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "splade or zipf or fuzz or edge" > gpurun_out/r2j/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r2j/pytest.log; [ $rc -eq 0 ] || exit $rc
bash tools/abl_libs.sh "libsparse_rx.so" "--workload c4 --steps 5" "--workload c5 --steps 5" > gpurun_out/r2j/abl_c4f.log 2>&1; cat gpurun_out/r2j/abl_c4f.log
timeout -k 10 300 python tools/stamp2_run.py --workload c4 > gpurun_out/r2j/stamp2_c4g.log 2>&1; tail -8 gpurun_out/r2j/stamp2_c4g.log
