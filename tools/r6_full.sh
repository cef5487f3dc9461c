# Round 6: the whole GPU suite, then the driver's bench line (python bench.py, no flags) — on one box.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6_gpu_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r6_gpu_tests.log
[ $rc = 0 ] && timeout -k 10 300 python bench.py > gpurun_out/r6_bench_c3.json 2> gpurun_out/r6_bench_c3.err && python3 -c "
import json; d=json.load(open('gpurun_out/r6_bench_c3.json')); print(d['value'], d['value_new_view'], d.get('value_moving_camera_2_in_flight'), d['roofline']['kernel_ms_avg'], d['cpu_baseline']['parity_vs_gpu'], d.get('host_surface',{}).get('sync'))"
