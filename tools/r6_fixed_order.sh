# Whole-frame parity of C2, C3 and C4 in the FIXED tile order (what a new view gets) on the round's final library: bench.py with the
# cpu leg on and the library's `auto` mode pinned → gpurun_out/r6_spec_{c2,c3,c4}_fixed_order_bench.json
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for w in c2 c3 c4; do
	LOL_BENCH_TILE_ORDER=auto LOL_BENCH_FRAMES_IN_FLIGHT_LEG=0 LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 timeout -k 10 300 \
		python3 bench.py --workload $w --steps 20 --warmup 5 2> gpurun_out/r6_fixed_$w.err | grep -o '^{.*' > gpurun_out/r6_spec_${w}_fixed_order_bench.json || { echo "$w failed"; tail -3 gpurun_out/r6_fixed_$w.err; exit 1; }
	python3 -c "import json; d=json.load(open('gpurun_out/r6_spec_${w}_fixed_order_bench.json')); print('$w', d['value'], d['tile_order'], d['config']['kernel_key'], d['cpu_baseline']['parity_vs_gpu']['pixels_compared'], d['cpu_baseline']['parity_vs_gpu']['pixels_differing'])" || exit 1
done
