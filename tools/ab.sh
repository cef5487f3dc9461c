export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h)
cd ${GRAFT_REPO_ROOT:?run on the GPU box}; mkdir -p gpurun_out
run() { echo -n "== $1: "; env $2 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 15 2>&1 | grep -o '"value": [0-9.]*\|"kernel_ms_avg": [0-9.]*' | tr '\n' ' '; echo; }
for rep in 1 2; do
run 8x8x4 "X=1"
run 8x8x1 "LOL_GPU_WAVE_SHAPE=8x8x1"
run 16x4x1 "LOL_GPU_WAVE_SHAPE=16x4x1"
run 32x2x1 "LOL_GPU_WAVE_SHAPE=32x2x1"
run 4x16x1 "LOL_GPU_WAVE_SHAPE=4x16x1"
done
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?run on the GPU box}; O=$R/gpurun_out/pmc_shape; mkdir -p $O; cd /tmp
LOL_GPU_WAVE_SHAPE=8x8x1 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/w1.log 2>&1
grep lol_render $O/w1/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | sort | uniq -c
