R=$GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests/test_gpu_grouped.py tests/test_gpu_ops.py tests/test_gpu_hpnn_chain.py tests/test_gpu_spectral64.py tests/test_gpu_spectral.py -x -v > $R/gpurun_out/r03_tests_d.log 2>&1; echo "rc=$?"; tail -15 $R/gpurun_out/r03_tests_d.log
