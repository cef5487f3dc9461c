# Round 6: fields of 3 ... 250 separate objects at 1080p, the library with the per-scene id policy (in-tree) against the one before it.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for lib in loltracer_amd/lib/liblol_gpu.so tools/ab/liblol_gpu_no_id_policy.so; do
	echo "== $lib"; LOL_GPU_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 500 python tools/flat_scene_ab.py --objects 2,3,4,16,64,150 2>/dev/null
done
