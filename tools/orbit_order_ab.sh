R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
for rep in 1 2; do
for kv in "LOL_BENCH_TILE_ORDER=lpt" "LOL_BENCH_TILE_ORDER=cols"; do
	env $kv LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload orbit 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kv', d['value'], d['roofline']['kernel_ms_avg'])"
done; done
for w in c3 c2 c4; do for kv in "LOL_BENCH_TILE_ORDER=lpt" "LOL_BENCH_TILE_ORDER=auto"; do
	env $kv LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $kv', d['value'], d['roofline']['kernel_ms_avg'], d['tile_order'])"
done; done
