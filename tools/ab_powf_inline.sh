export LOL_GPU_TUNING=1
# Round 5 experiment: powf_glibc inlined into the scene's kernel (no call left in it) against the out-of-line function, C3, still
# (the macro hunk in lol_kernel.h that read LOL_POWF_INLINE is not kept: +-0.1 %)
# camera (scheduled) and fixed column order, three pairs each, one box.
cd ${GRAFT_REPO_ROOT:?run on the GPU box}
for rep in 1 2 3; do for f in "" "-DLOL_POWF_INLINE=1"; do for o in lpt cols; do
LOL_GPU_RTC_FLAGS="$f" LOL_BENCH_TILE_ORDER=$o LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 LOL_BENCH_FRAMES_IN_FLIGHT_LEG=0 LOL_BENCH_SCHEDULING=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags=[$f] order=$o', d['value'], d['roofline']['kernel_ms_avg'])"
done; done; done
