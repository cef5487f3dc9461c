# Round profile: kernel-trace stats + PMC passes of the default bench (C3), outputs under gpurun_out/prof_final
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_final; mkdir -p $O; cd /tmp
B="python3 $R/bench.py --steps 20 --warmup 3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1 || exit 1
BP="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/write -- $BP > $O/write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $BP > $O/fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/sq -- $BP > $O/sq.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/sq2 -- $BP > $O/sq2.log 2>&1 || exit 1
grep -o '^{.*' $O/stats.log | tail -1 > $O/bench_line.json
find $O -name "*.csv" | wc -l
