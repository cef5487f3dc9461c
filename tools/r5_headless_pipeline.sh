cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_frames_in_flight.py tests/test_gpu_fuzz.py tests/test_cabi.py tests/test_gpu_boundary.py tests/test_headless_host.py -m gpu -x -q > gpurun_out/r5_t3.log 2>&1; tail -5 gpurun_out/r5_t3.log
H=loltracer_amd/lib/lol_headless; S=tests/golden/scenes/scene4.lol; mkdir -p gpurun_out/r5_fif
for cam in "--orbit" ""; do for flags in "" "--pipeline" "--pipeline-depth 3"; do
  n=$(echo "headless$cam$flags" | tr -d " -")
  timeout -k 10 120 $H 8 $S --size 3840x2160 --frames 120 $cam --wait-kernel $flags > gpurun_out/r5_fif/$n.log 2>&1
  echo "$n: $(grep Median gpurun_out/r5_fif/$n.log) mean30+: $(grep -o 'Frame [0-9]*: [0-9.]*ms' gpurun_out/r5_fif/$n.log | awk '{gsub("ms","",$3); if (NR>30) {s+=$3; n++}} END {printf "%.4f", s/n}')"
done; done
