export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h)
# sweep of the longest-tiles-first knobs on one box (re-sort period, base order): bash tools/lpt_sweep.sh
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
for base in rows cols; do for per in 1 2 4 8 32; do
	echo "base=$base resort=$per"
	LOL_GPU_LPT_BASE=$base LOL_GPU_LPT_RESORT=$per timeout -k 10 200 python tools/tile_order_ab.py --workloads c3,orbit,c4,c2 --kernels spec 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['workload'], 'rows', d['rows'], 'cols', d['cols'], 'lpt', d['lpt'], d['lpt_frame_equal'])"
done; done
