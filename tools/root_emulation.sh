# One GPU playing rank 0 of an 8-rank C4 run (bench.py --emulate-root-of 8): its share of the kernels + 1-rank RCCL gather + whole-frame
# assembly, with an equal share and with the 16/15 share the autotune picks.  Records -> gpurun_out/root_emulation/
R=${GRAFT_REPO_ROOT:?run on the GPU box}; O=$R/gpurun_out/root_emulation; mkdir -p $O; cd $R
TAG=${1:-r6}
python3 bench.py --emulate-root-of 8 --steps 40 --warmup 10 --no-cpu-baseline 2> /dev/null | grep -o '^{.*' > $O/${TAG}_root_emulation_equal.json
LOL_BENCH_ROOT_SHARE=16,15 python3 bench.py --emulate-root-of 8 --steps 40 --warmup 10 --no-cpu-baseline 2> /dev/null | grep -o '^{.*' > $O/${TAG}_root_emulation_bands_16_15.json
python3 - <<PY
import json, glob
for p in sorted(glob.glob("$O/${TAG}_*.json")):
    d = json.load(open(p)); print(p.split("/")[-1], d["value"], d["unit"], "root kernel", d.get("root_kernel_ms", d.get("root_kernel_ms_estimated")), "implied", d["implied_mpixels_per_s_if_root_is_the_critical_path"])
PY
