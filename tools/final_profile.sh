# Round profile on the GPU box: kernel-trace stats + PMC passes of the default bench (C3) for BOTH kernels, the
# other BASELINE workloads, and the summaries.  Everything lands under gpurun_out/prof_final; copy what is to be
# judged into profiles/ (see profiles/README.md).   usage: bash tools/final_profile.sh [round-tag, default r2]
TAG=${1:-r2}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_final; rm -rf $O; mkdir -p $O; cd /tmp
B="python3 $R/bench.py --steps 20 --warmup 5"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1 || exit 1
grep -o '^{.*' $O/stats.log | tail -1 > $O/${TAG}_spec_c3_bench.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_spec_c3_kernel_stats.csv
BP="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
pmc() { # name, counters...
	n=$1; shift
	timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- $BP > $O/$n.log 2>&1 || return 1
	cp $(find $O/$n -name "*counter_collection.csv" | head -1) $O/${TAG}_${PFX}_pmc_${n}_counter_collection.csv
}
PFX=spec
pmc write WRITE_SIZE GRBM_GUI_ACTIVE || exit 1
pmc fetch FETCH_SIZE || exit 1
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES || exit 1
pmc sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT || exit 1
python3 $R/tools/pmc_summary.py --kernel lol_render_spec --workload c3 --pixels 8294400 --out $O/pmc_traffic.json $O/${TAG}_spec_pmc_*_counter_collection.csv > $O/pmc_spec.txt || exit 1
export LOL_GPU_SPECIALIZE=0
PFX=interp
pmc write WRITE_SIZE GRBM_GUI_ACTIVE || exit 1
pmc fetch FETCH_SIZE || exit 1
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES || exit 1
pmc sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT || exit 1
python3 $R/tools/pmc_summary.py --merge --kernel render_interp --workload c3 --pixels 8294400 --out $O/pmc_traffic.json $O/${TAG}_interp_pmc_*_counter_collection.csv > $O/pmc_interp.txt || exit 1
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | grep -o '^{.*' > $O/${TAG}_interp_c3_bench.json
unset LOL_GPU_SPECIALIZE
cd $R
for w in c2 c4 orbit; do python3 bench.py --no-cpu-baseline --workload $w 2> /dev/null | grep -o '^{.*' > $O/${TAG}_spec_${w}_1gpu_bench.json; done
ls $O | head -50
