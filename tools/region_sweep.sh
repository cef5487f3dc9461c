export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h)
# the region whose pixels are dealt to waves by cost: sweep of its shape on one box.  bash tools/region_sweep.sh
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
for reg in 16x4 32x8 64x8 32x16 64x16 128x8 64x32 128x16 128x32 256x16 64x64; do
	echo "region $reg"
	LOL_GPU_REGION=$reg timeout -k 10 200 python tools/tile_order_ab.py --workloads c3,c4,c2 --kernels spec --frames 64 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['workload'], 'cols', d['cols'], 'lpt', d['lpt'], d['lpt_frame_equal'])"
done
