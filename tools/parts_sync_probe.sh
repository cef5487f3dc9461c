# Does a frame rendered as several parts on ONE device reach a host surface sooner through the unmodified per-frame protocol
# (part i copied while part i+1 renders)?  The C host, 4K, 60 frames, medians.  usage on the GPU box: bash tools/parts_sync_probe.sh
R=${GRAFT_REPO_ROOT:?run on the GPU box}; O=$R/gpurun_out/parts_sync; mkdir -p $O; cd $R
H=$R/loltracer_amd/lib/lol_headless; S=$R/tests/golden/scenes/scene4.lol
for cam in "" "--orbit"; do
	for flags in "" "--devices 0" "--devices 0 --parts-per-device 2" "--devices 0 --parts-per-device 3" "--devices 0 --parts-per-device 4" "--devices 0 --parts-per-device 8"; do
		n=$(echo "cam$cam$flags" | tr -d ' -')
		timeout -k 10 120 $H 8 $S --size 3840x2160 --frames 60 --wait-kernel $cam $flags > $O/$n.log 2>&1 || { echo "$n failed"; tail -3 $O/$n.log; exit 1; }
		echo "$n: $(grep Median $O/$n.log)"
	done
done
