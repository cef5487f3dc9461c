# Round 6: the whole GPU suite, then the driver's bench line (python bench.py, no flags), then the whole-frame parity records of the
# other configurations in the scheduled and the fixed-order mode — on one box.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gpu_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r6_gpu_tests.log
[ $rc = 0 ] || exit $rc
show() { python3 -c "
import json,sys; d=json.load(open('$1')); cb=d.get('cpu_baseline') or {}; print('$1', d['value'], d.get('value_new_view'), d['roofline']['kernel_ms_avg'], d.get('tile_order'), cb.get('parity_vs_gpu'), d.get('orbit_parity'))"; }
timeout -k 10 300 python bench.py > gpurun_out/r6_bench_c3.json 2> gpurun_out/r6_bench_c3.err && show gpurun_out/r6_bench_c3.json
for w in c2 c4 orbit; do
	LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 timeout -k 10 400 python bench.py --workload $w > gpurun_out/r6_spec_${w}_1gpu_bench.json 2> gpurun_out/r6_spec_${w}.err && show gpurun_out/r6_spec_${w}_1gpu_bench.json
done
for w in c2 c4; do
	LOL_BENCH_TILE_ORDER=auto LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 LOL_BENCH_FRAMES_IN_FLIGHT_LEG=0 timeout -k 10 400 python bench.py --workload $w > gpurun_out/r6_spec_${w}_fixed_order_bench.json 2> gpurun_out/r6_spec_${w}_fixed.err && show gpurun_out/r6_spec_${w}_fixed_order_bench.json
done
