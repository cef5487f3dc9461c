# A/B of library builds (tools/ab/*.so against the in-tree one) on the specialised kernel over C3, C2 and C4, six repetitions: value, new view, kernel ms.
export LOL_GPU_TUNING=1
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R; mkdir -p gpurun_out
for rep in 1 2 3 4 5 6; do for lib in loltracer_amd/lib/liblol_gpu.so tools/ab/*.so; do for w in c3 c2 c4; do
LOL_GPU_LIB=$R/$lib LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 LOL_BENCH_FRAMES_IN_FLIGHT_LEG=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 40 --workload $w 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$w', d['value'], d.get('value_new_view'), d['roofline']['kernel_ms_avg'])"
done; done; done
