R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_fif; mkdir -p $O; cd $R
rec() { grep -o '^{.*' | tail -1; }
for order in rows cols lpt; do for n in 1 2 3; do
	LOL_BENCH_TILE_ORDER=$order LOL_BENCH_FRAMES_IN_FLIGHT=$n LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload orbit 2> $O/orbit_o.err | rec > $O/orbit_${order}_$n.json || exit 1
	python3 -c "import json; d=json.load(open('$O/orbit_${order}_$n.json')); print('orbit', '$order', $n, 'in flight:', d['value'], 'Mpixels/s, kernel_ms_avg', d['roofline']['kernel_ms_avg'], d['tile_order'])"
done; done
H=$R/loltracer_amd/lib/lol_headless; S=$R/tests/golden/scenes/scene4.lol
for cam in "--orbit" ""; do for flags in "" "--pipeline" "--pipeline-depth 3" "--pipeline-depth 4"; do
	n=$(echo "headless$cam$flags" | tr -d ' -')
	timeout -k 10 120 $H 8 $S --size 3840x2160 --frames 120 $cam --wait-kernel $flags > $O/$n.log 2>&1 || exit 1
	echo "$n: $(grep -c Frame $O/$n.log) $(grep Median $O/$n.log)"
done; done
