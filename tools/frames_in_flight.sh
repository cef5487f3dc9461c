# Frames in flight (round 5, lol_gpu_set_frames_in_flight / --pipeline-depth): one GPU, one call, every A/B on the same box.
#  - bench.py's `frames_in_flight` leg on C3 and C2 (1 / 2 / 3 frames in flight; still and moving camera; frames compared)
#  - the orbit workload (config 5's per-rank work) with 1 / 2 / 3 / 4 frames in flight
#  - the C host through render_thread into a host surface: sequential, --pipeline, --pipeline-depth 3 / 4; orbit and still camera
# Everything lands under gpurun_out/r5_fif; copy what is to be judged into profiles/.   usage: bash tools/frames_in_flight.sh
R=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=$R/gpurun_out/r5_fif; mkdir -p $O; cd $R
rec() { grep -o '^{.*' | tail -1; }
for w in c3 c2; do
	LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload $w 2> $O/bench_$w.err | rec > $O/bench_$w.json || exit 1
	python3 -c "import json; d=json.load(open('$O/bench_$w.json')); print('$w value', d['value'], 'new_view', d['value_new_view'], json.dumps(d['frames_in_flight']))"
done
for rep in 1 2; do for n in 1 2 3 4; do
	LOL_BENCH_FRAMES_IN_FLIGHT=$n LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload orbit 2> $O/orbit_$n.err | rec > $O/orbit_${n}_rep$rep.json || exit 1
	python3 -c "import json; d=json.load(open('$O/orbit_${n}_rep$rep.json')); print('orbit', $n, 'in flight:', d['value'], 'Mpixels/s, kernel_ms_avg', d['roofline']['kernel_ms_avg'], d['tile_order'])"
done; done
H=$R/loltracer_amd/lib/lol_headless; S=$R/tests/golden/scenes/scene4.lol
for cam in "--orbit" ""; do for flags in "" "--pipeline" "--pipeline-depth 3" "--pipeline-depth 4"; do
	n=$(echo "headless$cam$flags" | tr -d ' -')
	timeout -k 10 120 $H 8 $S --size 3840x2160 --frames 120 $cam --wait-kernel $flags > $O/$n.log 2>&1 || exit 1
	echo "$n: $(grep Median $O/$n.log)"
done; done
ls $O
