R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/specsrc
for lib in loltracer_amd/lib/liblol_gpu.so tools/ab/prev.so; do n=$(basename $lib .so)
LOL_GPU_LIB=$R/$lib LOL_GPU_CACHE=0 LOL_GPU_DUMP_SPEC_SOURCE=$R/gpurun_out/specsrc/$n.c2.hip LOL_BENCH_HOST_SURFACE=0 python3 bench.py --no-cpu-baseline --steps 3 --workload c2 2>gpurun_out/specsrc/$n.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['config']['kernel_key'])"
python3 - <<PY
from loltracer_amd import gpu
import os
os.environ['LOL_GPU_LIB']='$R/$lib'
PY
done
ls -la gpurun_out/specsrc
