export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment")
# Round 5: PAIR records in the interpreter — an experiment that lives in commit 470c9f6 only (two spheres under one smooth union as ONE
# record; reverted: scene4 +4 %, scene.lol -4 %, the bar was +10 %).  This is the recipe that measured it: the in-tree library with and
# without pairs (same code, other lists) against the library built before the pair body existed (tools/ab/base.so), twice, one box.
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
for rep in 1 2; do for cfg in "loltracer_amd/lib/liblol_gpu.so 1" "loltracer_amd/lib/liblol_gpu.so 0" "tools/ab/base.so 1"; do set -- $cfg; for w in c3 c2 c4 orbit; do
LOL_GPU_LIB=$R/$1 LOL_GPU_INTERP_PAIRS=$2 LOL_GPU_SPECIALIZE=0 LOL_BENCH_HOST_SURFACE=0 LOL_BENCH_STARTUP=0 LOL_BENCH_FRAMES_IN_FLIGHT_LEG=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 pairs=$2', '$w', d['value'], d.get('value_new_view'), d['config']['kernel'])"
done; done; done
