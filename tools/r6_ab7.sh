cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
( LOL_GPU_LIB=$GRAFT_REPO_ROOT/tools/ab/liblol_gpu_divskip.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sdf.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r6_divskip_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_divskip_tests.log; tail -3 gpurun_out/r6_divskip_tests.log ) &&
bash tools/ab_spec_workloads.sh > gpurun_out/r6_ab_divskip.txt 2>&1; cat gpurun_out/r6_ab_divskip.txt
