export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h)
# Knobs tuned in rounds 2 - 3 on rectangle waves, swept again on the dealt waves of round 4 (C3, one call): Mpixels/s, kernel ms
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
run() { v=$(env "$@" LOL_GPU_CACHE_DIR= LOL_BENCH_STARTUP=0 LOL_BENCH_SCHEDULING=0 LOL_BENCH_HOST_SURFACE=0 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['value'], b['roofline']['kernel_ms_avg'])"); echo "$* : $v"; }
run X=0
for c in 0 1 2 4 6 10; do run LOL_GPU_CULL_COOLDOWN=$c; done
for w in 6,6 7,7 8,8 4,8; do run LOL_GPU_WAVES_PER_EU=$w; done
run LOL_GPU_SMIN_SAT=0
run LOL_GPU_SMIN_SAT=2
run LOL_GPU_SCHED=default
run X=0
run X=0
for n in 1 2 3 5 0; do run LOL_GPU_SAT_CULL_MIN_PRIMS=$n; done
for n in 1 2 4 8; do run LOL_GPU_CULL_CLUSTERS=$n; done
run LOL_GPU_NAN_FLAG=0
run X=0
