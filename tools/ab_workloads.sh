export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment")
# A/B of interpreter builds over the BASELINE workloads on one box: every library under tools/ab/ and the in-tree one, C2 / C3 / C4 on render_interp.
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
for rep in 1 2; do for lib in loltracer_amd/lib/liblol_gpu.so tools/ab/*.so; do for w in c2 c3 c4; do
LOL_GPU_LIB=$R/$lib LOL_GPU_SPECIALIZE=0 LOL_BENCH_HOST_SURFACE=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$w', d['value'], d['config']['kernel'])"
done; done; done
