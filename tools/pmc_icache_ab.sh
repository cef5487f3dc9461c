export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment")
# Instruction-cache view of interpreter builds (tools/ab/*.so and the in-tree library): SQC_ICACHE_* per render_interp dispatch on C3.
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?run on the GPU box}; O=$R/gpurun_out/pmc_icache; rm -rf "$O"; mkdir -p $O; cd /tmp
export LOL_GPU_SPECIALIZE=0 LOL_BENCH_HOST_SURFACE=0
for lib in loltracer_amd/lib/liblol_gpu.so tools/ab/liblol_gpu_NO_RUNS.so; do
	n=$(basename $lib .so)
	export LOL_GPU_LIB=$R/$lib
	timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/$n -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/$n.log 2>&1 || exit 1
	python3 - "$n" $(find $O/$n -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    if "render_interp" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}, "dispatches", len(next(iter(acc.values()))))
PY
done
