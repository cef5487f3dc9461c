export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h)
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
run() { echo "$1"; env $1 LOL_GPU_LPT_RESORT=4 timeout -k 10 200 python tools/tile_order_ab.py --workloads c3,orbit,c4,c2,band --kernels spec 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['workload'], 'rows', d['rows'], 'cols', d['cols'], 'lpt', d['lpt'], d['lpt_frame_equal'])"; }
run "X=1"
run "LOL_GPU_RTC_FLAGS=-DLOL_LPT_COST_CYCLES=1 LOL_GPU_LPT_SHIFT=0"
run "LOL_GPU_RTC_FLAGS=-DLOL_LPT_COST_CYCLES=1 LOL_GPU_LPT_SHIFT=1"
run "LOL_GPU_LPT_SHIFT=1"
