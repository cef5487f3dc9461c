export LOL_GPU_TUNING=1   # the library honours its A/B switches only beside this (include/lol_gpu.h)
# hipRTC option sweep for the specialised kernel (C3 and C2), one process per case (an LLVM that does not know an -mllvm option ends
# the process): usage on the GPU box: bash tools/rtc_flag_sweep.sh   → one line per case: flags | C3 Mpixels/s | C2 Mpixels/s
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
run() { # $1 = LOL_GPU_SCHED value ("" = library default), rest = extra flags
	s=$1; shift
	for w in c3 c2; do
		v=$(LOL_GPU_SCHED=$s LOL_GPU_RTC_FLAGS="$*" LOL_GPU_CACHE=0 LOL_BENCH_HOST_SURFACE=0 timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 30 --workload $w 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'] if d['config']['kernel']=='lol_render_spec' else 'not-spec')" 2>/dev/null || echo fail)
		printf "%s " "$v"
	done
	echo "| sched=${s:-lib} $*"
}
if [ "$1" = ifcvt ]; then for rep in 1 2; do
run ""
run "" -mllvm -amdgpu-early-ifcvt=1
run "" -mllvm -simplifycfg-branch-fold-threshold=4
run "" -mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1
run "" -mllvm -amdgpu-set-wave-priority=1
run "" -mllvm -phi-node-folding-threshold=8 -mllvm -amdgpu-early-ifcvt=1
run "" -mllvm -enable-shrink-wrap=0
run "" -mllvm -amdgpu-atomic-optimizer-strategy=None
done; exit 0; fi
if [ "$1" = phi ]; then for rep in 1 2 3; do
run ""
for t in 3 4 6 8 12 16 32 64; do run "" -mllvm -phi-node-folding-threshold=$t; done
done; exit 0; fi
if [ "$1" = more ]; then for rep in 1 2; do
run ""
run "" -funroll-loops
run "" -mllvm -unroll-count=2
run "" -mllvm -inline-threshold=10000
run "" -mllvm -enable-gvn-hoist=1
run "" -mllvm -enable-gvn-sink=1
run "" -mllvm -enable-unroll-and-jam=1
run "" -mllvm -two-entry-phi-node-folding-threshold=16
run "" -mllvm -phi-node-folding-threshold=8
run "" -mllvm -amdgpu-vgpr-index-mode=1
run "" -mllvm -enable-loop-versioning-licm=1
run "" -mllvm -speculative-execution-max-speculation-cost=32
done; exit 0; fi
for rep in 1 2; do
run ""
run "" -mllvm -amdgpu-use-aa-in-codegen=0
run "" -mllvm -misched-cluster=0
run "" -mllvm -amdgpu-early-inline-all=1
run "" -fno-unroll-loops
run "" -mllvm -amdgpu-promote-alloca-to-vector-limit=0
run "" -mllvm -greedy-regclass-priority-trumps-globalness=1
run "" -mllvm -amdgpu-schedule-relaxed-occupancy=1
run default -mllvm -amdgpu-sched-strategy=iterative-ilp
run default -mllvm -amdgpu-sched-strategy=iterative-maxocc
run "" -mllvm -enable-machine-outliner=never
run "" -O2
done
