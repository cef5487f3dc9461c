# Round 6, last call: the profile recipe on the final library, the whole GPU suite and smoke() as the driver runs them, then soaks.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
bash tools/final_profile.sh r6 > gpurun_out/r6_final_profile.log 2>&1; echo "profile rc=$?"; tail -2 gpurun_out/r6_final_profile.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gpu_tests_final.log 2>&1; echo "suite rc=$?"; tail -2 gpurun_out/r6_gpu_tests_final.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r6_smoke.log
run() { n=$1; seed=$2; shift 2; tag=$(echo "$*" | tr ' ' '_'); f=gpurun_out/r6_soak_last_${tag:-plain}_${n}_scenes.log
	timeout -k 10 600 python tests/tools/soak.py $n $seed "$@" > $f 2>&1; echo "$f rc=$? $(tail -1 $f)"; }
run 250 62001
run 150 62002 stress
run 120 62003 still
run 120 62004 flight
