# Round 5: the N>1 bench with the kernels of consecutive frames on one stream (LOL_BENCH_KERNEL_STREAMS=1, as before) against one
# stream per slot of the gather pipeline — on ONE GPU: rank 0 of an 8-rank C4 run emulated (its bands + 1-rank RCCL gather + the
# whole-frame assembly), three times each, and the 2-rank rehearsal over gloo with the assembled frame checked.
R=${GRAFT_REPO_ROOT:?run on the GPU box}; cd $R
for rep in 1 2 3; do for k in 1 slots; do for share in equal 16,15; do
LOL_BENCH_KERNEL_STREAMS=$k LOL_BENCH_ROOT_SHARE=$share timeout -k 10 200 python3 bench.py --emulate-root-of 8 --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('emulate root of 8, kernel streams=$k share=$share:', d['value'], 'ms/frame, root kernel', d['root_kernel_ms'], d['config']['kernel_streams'])"
done; done; done
for k in 1 slots; do
LOL_BENCH_KERNEL_STREAMS=$k LOL_BENCH_REHEARSE=1 LOL_BENCH_CHECK=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks over gloo, kernel streams=$k:', d['value'], 'Mpixels/s, frame equal', d['frame_equal_to_single_launch'], d['kernel_ms'])"
done
