# Rehearsal of the N>1 bench on ONE GPU (VERDICT round 3, next #1e): worlds 2, 4 and 5 over gloo with every rank on device 0 —
# partition, autotuned root share, pipelined gather, assembly, the default frame check and the per-rank record are the real
# ones; only the transport (gloo instead of RCCL over xGMI) and the shared device differ from the 8-GPU run.  World 8 is
# not rehearsed: a GPU box admits at most 6 processes on its card at once (the pool's process guard), and the launcher side holds one more handle: 6 ranks were killed by it.  Also the orbit over
# 4 ranks and the in-process transport.  Records land under gpurun_out/rehearse; copy what is to be judged into profiles/.
# usage (on the GPU box): bash tools/rehearse_worlds.sh [tag, default r4]
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=$R/gpurun_out/rehearse; mkdir -p $O; cd $R
rec() { grep -o '^{.*' | tail -1; }
export LOL_BENCH_REHEARSE=1
for n in 2 4 5; do
	echo "world $n"
	timeout -k 10 400 python3 bench.py --gpus $n --steps 10 --warmup 2 2> $O/world$n.err | rec > $O/${TAG}_rehearse_${n}ranks_gloo.json || { tail -5 $O/world$n.err; exit 1; }
	grep "\[check\]" $O/world$n.err
done
echo "orbit over 4 ranks"
timeout -k 10 400 python3 bench.py --gpus 4 --workload orbit --warmup 1 2> $O/orbit4.err | rec > $O/${TAG}_rehearse_orbit_4ranks_gloo.json || { tail -5 $O/orbit4.err; exit 1; }
grep "\[check\]" $O/orbit4.err
unset LOL_BENCH_REHEARSE
echo "cabi"
timeout -k 10 300 python3 bench.py --transport cabi --gpus 1 --workload c4 --steps 20 --warmup 3 2> $O/cabi.err | rec > $O/${TAG}_cabi_c4_1gpu.json || { tail -5 $O/cabi.err; exit 1; }
grep "\[check\]" $O/cabi.err
python3 - <<PY
import json, glob
for p in sorted(glob.glob("$O/${TAG}_*.json")):
    d = json.load(open(p))
    print(p.split("/")[-1], d["value"], d["unit"], "equal:", d.get("frame_equal_to_single_launch", d.get("frames_equal_to_rank0_render")),
          "kernel_ms:", d.get("kernel_ms"), "gather_exposed_ms:", d.get("gather_exposed_ms", d.get("gather_exposed_ms_estimated")))
PY
