export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h)
# Pixel footprint of a wave of the specialised kernel (LOL_GPU_WAVE_SHAPE=WxHxWAVES) over the BASELINE workloads, one box:
# usage on the GPU box: bash tools/wave_shape_ab.sh   → one line per shape: C2 C3 C4 orbit Mpixels/s
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
for rep in 1 2; do for sh in 16x4x1 8x8x1 4x16x1 2x32x1 1x64x1; do
	for w in c2 c3 c4 orbit; do
		v=$(LOL_GPU_WAVE_SHAPE=$sh LOL_BENCH_HOST_SURFACE=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])" 2>/dev/null || echo fail)
		printf "%s " "$v"
	done; echo "| $sh"
done; done
