export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment")
# PMC passes of the interpreter kernel on C3 (LOL_GPU_SPECIALIZE=0 makes bench.py time render_interp); outputs under gpurun_out/<name>
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?run on the GPU box}; O=$R/gpurun_out/${1:-pmc_interp2}; mkdir -p $O; cd /tmp
export LOL_GPU_SPECIALIZE=0
BP="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/sq -- $BP > $O/sq.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -- $BP > $O/sq2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for d in ("sq", "sq2"):
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "render_interp" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(d, k, sum(v) / len(v), len(v))
PY
