# A/B in one call on one box: interleaved bench runs with environment variants.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo -n "== $1: "; env $2 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 15 2>&1 | grep -o '"value": [0-9.]*\|"kernel_ms_avg": [0-9.]*' | tr '\n' ' '; echo; }
for rep in 1 2; do
run skip_on "X=1"
run skip_off "LOL_GPU_MISS_SKIP=0"
done
