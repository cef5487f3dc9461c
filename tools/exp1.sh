cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "== $1 | $2"; LOL_GPU_RTC_FLAGS="$2" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 2>&1 | grep -o '"value": [0-9.]*\|"kernel_ms_avg": [0-9.]*' | tr '\n' ' '; echo; }
run base ""
run noslp "-fno-slp-vectorize"
run O2 "-O2"
run noslp_unroll "-fno-slp-vectorize -fno-unroll-loops"
run waves4 "-mllvm -amdgpu-waves-per-eu=4"  
run maxocc "-mllvm --amdgpu-schedule-metric-bias=100"
