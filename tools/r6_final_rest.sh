# Round 6, after tools/final_profile.sh r6: the whole GPU suite and smoke() as the driver runs them, then soaks (tools/r6_final.sh without the profile).
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gpu_tests_final.log 2>&1; echo "suite rc=$?"; tail -2 gpurun_out/r6_gpu_tests_final.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r6_smoke.log
run() { n=$1; seed=$2; shift 2; tag=$(echo "$*" | tr ' ' '_'); f=gpurun_out/r6_soak_last_${tag:-plain}_${n}_scenes.log
	timeout -k 10 600 python tests/tools/soak.py $n $seed "$@" > $f 2>&1; echo "$f rc=$? $(tail -1 $f)"; }
run 250 63001
run 150 63002 stress
run 120 63003 still
run 120 63004 flight
