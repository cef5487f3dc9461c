export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h)
# Round 6: A/B of the march / shadow loop forms of lol_kernel.h (LOL_X_* / LOL_COUNT_STEPS, hipRTC -D flags) on one box.
# usage on the GPU box: bash tools/r6_ab.sh [workload] "flags A" "flags B" ...   ("" = the default kernel)
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R; W=$1; shift
for rep in 1 2 3; do for f in "$@"; do
	extra="--no-cpu-baseline"; [ $rep = 1 ] && extra=""
	LOL_GPU_RTC_FLAGS="$f" LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 timeout -k 10 300 python3 bench.py $extra --steps 40 --workload $W 2>/dev/null |
		python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W flags=[$f]', d['value'], d.get('value_new_view'), d.get('value_moving_camera_2_in_flight'), d['roofline']['kernel_ms_avg'], d['config']['kernel'], (d.get('cpu_baseline') or {}).get('parity_vs_gpu'))"
done; done
