export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment")
# Round profile on the GPU box: kernel-trace stats + PMC passes of the default bench (C3) for BOTH kernels, the
# other BASELINE workloads, and the summaries.  Everything lands under gpurun_out/prof_final; copy what is to be
# judged into profiles/ (see profiles/README.md).   usage: bash tools/final_profile.sh [round-tag, default r6]
TAG=${1:-r6}
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}; O=$R/gpurun_out/prof_final
rm -rf "$O"; mkdir -p "$O"; cd /tmp
export LOL_BENCH_HOST_SURFACE=0                     # the PMC passes time kernels; the host-surface leg has its own record
export LOL_BENCH_FRAMES_IN_FLIGHT_LEG=0             # ... and so has the frames-in-flight leg (round 5): its frames are other frames (moving camera, fixed order)
B="python3 $R/bench.py --steps 20 --warmup 5"
# The scene kernels are compiled by whichever hipRTC the process has loaded, and the one a process gets under rocprofv3 is not the
# one a plain `python bench.py` gets (torch brings its own; the profiler preloads the system's): the same source gave three
# different code objects that way (keys a9cd… / f6c3… / 254a…).  So the kernels the profiler times are the ones a plain run
# compiles: one plain run first fills the on-disk code-object cache, the profiled runs load from it, and the recorded
# kernel_key is the one the driver's own `python bench.py` will see.
export LOL_GPU_CACHE_DIR=$O/code_cache LOL_GPU_CACHE_ANY_COMPILER=1      # a cache of this recipe's own, shared by all its runs
# (the start-up leg compiles the scene with the disk cache switched off and the process then keeps THAT code object: the plain
#  run would store nothing, a profiled run would use the other compiler's — so all of these go without that leg, and a plain run
#  at the end records it)
export LOL_BENCH_STARTUP=0
python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/plain_first.json 2> /dev/null || exit 1
LOL_GPU_SPECIALIZE=0 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
keyof() { python3 -c "import json,sys; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['config']['kernel_key'])" $1; }
BP="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
export LOL_BENCH_SCHEDULING=0        # the counter passes average over dispatches: no fixed-order frames of the `scheduling` leg among them
pmc() { # name, counters...  → one rocprofv3 pass in a directory of its own ($PFX = which kernel the bench times)
	n=$1; shift
	timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/${PFX}_$n -- $BP > $O/${PFX}_$n.log 2>&1 || return 1
	[ -z "$WANT" ] || grep -q "\"kernel_key\": \"$WANT\"" $O/${PFX}_$n.log || { echo "pass $n ran another kernel than $WANT"; return 1; }
	cp "$(find $O/${PFX}_$n -name "*counter_collection.csv" | head -1)" $O/${TAG}_${PFX}_pmc_${n}_counter_collection.csv
}
passes() {
	pmc write WRITE_SIZE GRBM_GUI_ACTIVE || return 1
	pmc fetch FETCH_SIZE || return 1
	pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES || return 1
	pmc sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT || return 1
	# third pass (round 3): where the cycles between 2.0 and 2.7 per VALU instruction go — transcendental issue, scalar-unit
	# busy time, LDS waits, instruction fetch
	pmc sq3 SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES || return 1
}
key() { $BP 2> /dev/null | python3 -c "import json,sys; c=json.loads(sys.stdin.read())['config']; print(c['kernel'], c['kernel_key'])"; }
PFX=spec; WANT=$(keyof $O/plain_first.json)
passes || exit 1
set -- $(key); [ "$1" = lol_render_spec ] || { echo "expected lol_render_spec, bench timed $1"; exit 1; }
python3 $R/tools/pmc_summary.py --kernel lol_render_spec --kernel-key $2 --min-dispatches 100 --workload c3 --pixels 8294400 --out $O/pmc_traffic.json $O/${TAG}_spec_pmc_*_counter_collection.csv > $O/pmc_spec.txt || exit 1
export LOL_GPU_SPECIALIZE=0
PFX=interp; WANT=
passes || exit 1
set -- $(key); [ "$1" = render_interp ] || { echo "expected render_interp, bench timed $1"; exit 1; }
python3 $R/tools/pmc_summary.py --merge --kernel render_interp --kernel-key $2 --min-dispatches 100 --workload c3 --pixels 8294400 --out $O/pmc_traffic.json $O/${TAG}_interp_pmc_*_counter_collection.csv > $O/pmc_interp.txt || exit 1
# the counter summary of THIS run is what the records below quote (`roofline.traffic`, `valu.issue_frac`): bench.py reads
# profiles/pmc_traffic.json, so it goes there now — on the box; the same file is copied into the repository afterwards
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
LOL_BENCH_SCHEDULING=1 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | grep -o '^{.*' > $O/${TAG}_interp_c3_bench.json
unset LOL_GPU_SPECIALIZE
# kernel-trace statistics of the bench's own frames only (no `scheduling` / host-surface legs: their frames are other frames),
# so that the average can be held against the HIP-event average in the record
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1 || exit 1
grep -o '^{.*' $O/stats.log | tail -1 > $O/${TAG}_spec_c3_bench.json
[ "$(keyof $O/plain_first.json)" = "$(keyof $O/${TAG}_spec_c3_bench.json)" ] || { echo "the profiled run compiled its own kernel ($(keyof $O/${TAG}_spec_c3_bench.json)), not the plain run's ($(keyof $O/plain_first.json))"; exit 1; }
cp "$(find $O/stats -name "*kernel_stats.csv" | head -1)" $O/${TAG}_spec_c3_kernel_stats.csv
# the passes of the two kernels must not be copies of one another (round 2's were: one output directory for both)
for n in write fetch sq sq2 sq3; do
	if cmp -s $O/${TAG}_spec_pmc_${n}_counter_collection.csv $O/${TAG}_interp_pmc_${n}_counter_collection.csv; then echo "spec and interp $n CSVs are identical"; exit 1; fi
done
cd $R
unset LOL_BENCH_STARTUP LOL_BENCH_SCHEDULING LOL_BENCH_HOST_SURFACE LOL_BENCH_FRAMES_IN_FLIGHT_LEG
python3 bench.py 2> /dev/null | grep -o '^{.*' > $O/${TAG}_spec_c3_plain_bench.json
for w in c2 c4 orbit; do python3 bench.py --workload $w 2> /dev/null | grep -o '^{.*' > $O/${TAG}_spec_${w}_1gpu_bench.json; done
ls $O | grep -v "^spec_\|^interp_\|^stats$"
