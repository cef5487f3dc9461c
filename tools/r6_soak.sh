# Round 6: long soaks on the round's final library — random scenes x both kernels against the oracle (pixels, colours, ids, distances,
# step counts): ordinary, stress (crowds), one-k unions, the repeated view (fifth frame: dealt + longest first), two frames in flight.
# usage on the GPU box: bash tools/r6_soak.sh a|b [seed base, default 63000]     (two calls: each stays inside one gpurun limit)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out; B=${2:-63000}
run() { n=$1; seed=$2; shift 2; tag=$(echo "$*" | tr ' ' '_'); f=gpurun_out/r6_soak_long_${tag:-plain}_${n}_scenes.log
	timeout -k 10 1000 python tests/tools/soak.py $n $seed "$@" > $f 2>&1; echo "$f rc=$? $(tail -1 $f)"; }
if [ "$1" = a ]; then
run 500 $((B+1))
run 300 $((B+2)) stress
run 150 $((B+3)) onek
else
run 250 $((B+4)) still
run 150 $((B+5)) still stress
run 250 $((B+6)) flight
run 120 $((B+7)) flight stress
fi
