# Round 6, last call: the profile recipe on the final library, then the whole GPU suite and smoke() as the driver runs them.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
bash tools/final_profile.sh r6 > gpurun_out/r6_final_profile.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/r6_final_profile.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gpu_tests_final.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r6_gpu_tests_final.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r6_smoke.log
