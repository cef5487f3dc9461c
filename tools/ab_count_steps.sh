export LOL_GPU_TUNING=1
# Round 5 experiment (the hunk that read LOL_COUNT_STEPS in lol_kernel.h is not kept): the per-lane step counters cost 0.7 % of a C3 frame.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for f in "" "-DLOL_COUNT_STEPS=0"; do for o in cols; do
LOL_GPU_RTC_FLAGS="$f" LOL_BENCH_TILE_ORDER=$o LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 LOL_BENCH_FRAMES_IN_FLIGHT_LEG=0 LOL_BENCH_SCHEDULING=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags=[$f] order=$o', d['value'], d['roofline']['kernel_ms_avg'])"
done; done; done
