# One-GPU measurements behind the 8-GPU run: the root's un-interleave both ways, the cadence of
# rank 0 emulated on one device, the in-process transport, a 2-rank rehearsal of the autotuned split, and the C host.
# Everything lands under gpurun_out/multi; copy what is to be judged into profiles/.   usage: bash tools/multi_readiness.sh
R=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=$R/gpurun_out/multi; mkdir -p $O; cd $R
rec() { grep -o '^{.*' | tail -1; }
timeout -k 10 200 python3 tools/assemble_ab.py > $O/assemble_ab.json 2> $O/assemble_ab.err || exit 1
for share in 16,15 12,10; do
	n=$(echo $share | tr , _)
	LOL_BENCH_ROOT_SHARE=$share timeout -k 10 200 python3 bench.py --emulate-root-of 8 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/emu_$n.err | rec > $O/root_emulation_bands_$n.json || exit 1
done
LOL_BENCH_ROOT_SHARE=equal timeout -k 10 200 python3 bench.py --emulate-root-of 8 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/emu_eq.err | rec > $O/root_emulation_equal.json || exit 1
LOL_BENCH_ROOT_SHARE=equal LOL_BENCH_ASSEMBLE=torch timeout -k 10 200 python3 bench.py --emulate-root-of 8 --steps 40 --warmup 5 --no-cpu-baseline 2> $O/emu_eq_torch.err | rec > $O/root_emulation_equal_torch_assembly.json || exit 1
LOL_BENCH_CHECK=1 timeout -k 10 200 python3 bench.py --transport cabi --gpus 1 --workload c4 --steps 20 --warmup 3 2> $O/cabi.err | rec > $O/cabi_c4_1gpu.json || exit 1
LOL_BENCH_CHECK=1 LOL_BENCH_PARTS_PER_DEVICE=8 timeout -k 10 200 python3 bench.py --transport cabi --gpus 1 --workload c4 --steps 20 --warmup 3 2> $O/cabi8.err | rec > $O/cabi_c4_1gpu_8parts.json || exit 1
LOL_BENCH_REHEARSE=1 LOL_BENCH_CHECK=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 2 2> $O/rehearse2.err | rec > $O/rehearse_2ranks_gloo.json || exit 1
H=$R/loltracer_amd/lib/lol_headless; S=$R/tests/golden/scenes/scene4.lol
for flags in "" "--pipeline" "--devices 0"; do
	n=$(echo "orbit$flags" | tr -d ' -')
	# (--wait-kernel: 60 frames take less time than the scene compiler; without it a cold run measures the interpreter)
	timeout -k 10 120 $H 8 $S --size 3840x2160 --frames 60 --orbit --wait-kernel $flags > $O/headless_$n.log 2>&1 || exit 1
	grep Median $O/headless_$n.log
done
# ... and the same host with a camera that stands still (the scheduled frame, LABNOTES.md §3.9)
for flags in "" "--pipeline"; do
	n=$(echo "still$flags" | tr -d ' -')
	timeout -k 10 120 $H 8 $S --size 3840x2160 --frames 60 --wait-kernel $flags > $O/headless_$n.log 2>&1 || exit 1
	grep Median $O/headless_$n.log
done
ls $O
