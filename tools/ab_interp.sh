export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment")
# A/B of interpreter builds on one box: every library under tools/ab/ (and the in-tree one) times C3 on render_interp.
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
for rep in 1 2; do
for lib in loltracer_amd/lib/liblol_gpu.so tools/ab/*.so; do
	LOL_GPU_LIB=$R/$lib LOL_GPU_SPECIALIZE=0 LOL_BENCH_HOST_SURFACE=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['roofline']['kernel_ms_avg'], d['config']['kernel'])"
done; done
