# Round 6: soaks on the round's final library — random scenes x both kernels against the oracle (pixels, colours, ids, distances,
# step counts): ordinary, stress (crowds), one-k unions, the repeated view (fifth frame: dealt + longest first), two frames in flight.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
run() { n=$1; seed=$2; shift 2; tag=$(echo "$*" | tr ' ' '_'); f=gpurun_out/r6_soak_final_${tag:-plain}_${n}_scenes.log
	timeout -k 10 1000 python tests/tools/soak.py $n $seed "$@" > $f 2>&1; echo "$f rc=$? $(tail -1 $f)"; }
run 400 61001
run 250 61002 stress
run 150 61003 onek
run 200 61004 still
run 150 61005 still stress
run 200 61006 flight
run 120 61007 flight stress
