export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment")
# The specialised kernel's GENERATED source of two library builds (in-tree and tools/ab/*.so), for one workload: to find out whether a
# change of speed comes from the generated part or from lol_kernel.h (compile both offline with hipcc -S and diff: LABNOTES.md §3.1).
# usage on the GPU box: bash tools/dump_spec_ab.sh [c2|c3]   → gpurun_out/specsrc/<lib>.<workload>.hip
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R; W=${1:-c2}; mkdir -p gpurun_out/specsrc
for lib in loltracer_amd/lib/liblol_gpu.so tools/ab/*.so; do n=$(basename $lib .so)
	LOL_GPU_LIB=$R/$lib LOL_GPU_CACHE_DIR= LOL_GPU_DUMP_SPEC_SOURCE=$R/gpurun_out/specsrc/$n.$W.hip LOL_BENCH_HOST_SURFACE=0 \
		python3 bench.py --no-cpu-baseline --steps 3 --workload $W 2> gpurun_out/specsrc/$n.err |
		python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['config']['kernel_key'])"
done
