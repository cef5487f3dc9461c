export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h)
# A/B of gamma + quantisation through the proven table (default) against the powf route (LOL_GPU_GAMMA_TABLE=0); same box, same call.
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
for rep in 1 2; do for wl in c3 c2 c4; do for g in 1 0; do
	v=$(LOL_GPU_GAMMA_TABLE=$g LOL_BENCH_STARTUP=0 LOL_BENCH_SCHEDULING=0 LOL_BENCH_HOST_SURFACE=0 python3 bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['value'], b['roofline']['kernel_ms_avg'], b.get('frame_equal_to_single_launch'), b['kernels'])")
	echo "$wl gamma_table=$g $v"
done; done; done
