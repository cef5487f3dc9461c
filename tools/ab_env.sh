export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment")
# A/B of environment switches on one box: `bash tools/ab_env.sh "A=1" "A=0" ...` times C3 (default kernel) under each setting, twice.
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
for rep in 1 2; do for kv in "$@"; do
	env $kv LOL_BENCH_HOST_SURFACE=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kv', d['value'], d['roofline']['kernel_ms_avg'], d['kernels'].get('render_interp',{}).get('mpixels_per_s'), d['kernels'].get('render_interp',{}).get('frame_equal_to_spec'))"
done; done
